import os, sys
sys.path.insert(0, "/root/repo")
import torch
from siammask_amd import _lib, synth
from siammask_amd.custom import build
B = 8
m = build("sharp", dtype="f16", max_batch=B, graph=False)
m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
m = m.eval().cuda()
z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=3)).cuda()
x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=3)).cuda()
m.template(z)
pass
for i in range(3):
    m.track_mask(x)
torch.cuda.synchronize()
