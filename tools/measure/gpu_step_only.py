#!/usr/bin/env python
"""200 fused B=8 steps and nothing else (no result bookkeeping): for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from siammask_amd import synth
from siammask_amd.custom import build

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = build("sharp", dtype="f16", max_batch=B, graph=True)
m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
m = m.eval().cuda()
z = torch.from_numpy(synth.image_batch(B, 127, stream0=3)).cuda()
x = torch.from_numpy(synth.image_batch(B, 255, stream0=3)).cuda()
twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
m.template(z)
import time
for _ in range(20):
    m.track_step(x, twh, refine=True, stage=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    m.track_step(x, twh, refine=True, stage=False)
torch.cuda.synchronize()
print("B=%d %.4f ms/step" % (B, (time.perf_counter() - t0) / 200 * 1e3), m.seq_status(), {k: os.environ.get(k) for k in
      ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "HIP_FORCE_DEV_KERNARG", "DEBUG_HIP_GRAPH_KERNARG")})
