#!/usr/bin/env python
"""A/B in one process: sharp fp16 fused step with the persistent per-XCD sequences on / off.
Prints ms/step for each setting (graph replay, 100 steps after warm-up), the max difference of the outputs, and the
sequence status flag."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from siammask_amd import _lib, synth
from siammask_amd.custom import build


def run(B, seq, steps=100, extra=None):
    _lib.tune(seq=seq, **(extra or {}))
    m = build("sharp", dtype="f16", max_batch=B, graph=True)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    m = m.eval().cuda()
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=3)).cuda()
    x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=3)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
    m.template(z)
    for _ in range(10):
        o = m.track_step(x, twh, refine=True, stage=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        o = m.track_step(x, twh, refine=True, stage=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    st = m.seq_status()
    out = {k: v.clone() for k, v in o.items() if v is not None}
    del m
    return dt, st, out


def main():
    for B in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "8,16,64").split(",")]:
        d0, s0, o0 = run(B, 0)
        d1, s1, o1 = run(B, 1)
        d0b, _, _ = run(B, 0)
        d1b, _, _ = run(B, 1)
        err = {k: float((o0[k].double() - o1[k].double()).abs().max() / (o0[k].double().abs().max() + 1e-30)) for k in o0}
        print("B=%d  seq off %.4f / %.4f ms   seq on %.4f / %.4f ms   x%.3f   status %s   rel diff %s" % (
            B, d0, d0b, d1, d1b, min(d0, d0b) / min(d1, d1b), s1, {k: "%.1e" % v for k, v in err.items()}), flush=True)


if __name__ == "__main__":
    main()
