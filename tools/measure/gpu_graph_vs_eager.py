#!/usr/bin/env python
"""sharp fp16 fused step: hipGraph replay vs eager launches (24 kernels per step at B = 8: is the graph still needed, and
does hipGraphLaunch add device work of its own -- rocprofv3 shows ~3 __amd_rocclr_copyBuffer per step)?  ms/step, one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from siammask_amd import synth
from siammask_amd.custom import build


def run(B, graph, steps=150):
    m = build("sharp", dtype="f16", max_batch=B, graph=graph)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    m = m.eval().cuda()
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=3)).cuda()
    x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=3)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
    m.template(z)
    for _ in range(10):
        m.track_step(x, twh, refine=True, stage=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.track_step(x, twh, refine=True, stage=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    del m
    return (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3


for B in (8, 1):
    row = []
    for g in (True, False, True, False):
        ms, host = run(B, g)
        row.append("graph=%d %.4f (host %.3f)" % (g, ms, host))
    print("B=%d ms/step: %s" % (B, " | ".join(row)), flush=True)
