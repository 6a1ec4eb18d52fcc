#!/usr/bin/env python
"""Experiment: B streams as N independent chains (N contexts x B/N streams) on N HIP streams,
vs one chain of B.  Measures whether concurrent chains fill the launch/tail bubbles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from siammask_amd import synth
from siammask_amd.custom import build

def run(total_b, chains, steps=60, dtype="f16"):
    dev = torch.device("cuda", 0)
    b = total_b // chains
    ms, xs, zs, tw, st = [], [], [], [], []
    for c in range(chains):
        m = build("sharp", dtype=dtype, max_batch=b, graph=True)
        m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
        m = m.eval().to(dev)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            m.template(torch.from_numpy(synth.image_batch(b, 127, stream0=c * b)).to(dev))
        ms.append(m); st.append(s)
        xs.append(torch.from_numpy(synth.image_batch(b, 255, stream0=1000 + c * b)).to(dev))
        tw.append(torch.full((b, 2), 70.0, device=dev))
    torch.cuda.synchronize()
    def step():
        for c in range(chains):
            with torch.cuda.stream(st[c]):
                ms[c].track_step(xs[c], tw[c], refine=True)
    for _ in range(15): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt * 1e3, total_b / dt

for total_b, chains in ((8, 1), (8, 2), (8, 4), (16, 2), (64, 1), (64, 2), (1, 1)):
    msps, fps = run(total_b, chains)
    print("B=%d chains=%d: %.3f ms/step  %.0f fps" % (total_b, chains, msps, fps), flush=True)
