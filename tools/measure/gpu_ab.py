#!/usr/bin/env python
"""Generic in-process A/B of one smk_tune knob on the heavy conv shapes + the end-to-end step.
usage: gpu_ab.py <knob> <v0,v1,...>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from siammask_amd import _lib, ops, synth
from siammask_amd.custom import build
knob, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
SHAPES = {"l3.0.ds": (512, 31, 1024, 3, 1, 1, 1), "l3.c2": (256, 31, 256, 3, 1, 2, 2), "l3.c3": (256, 31, 1024, 1, 1, 0, 1),
          "l3.c1": (1024, 31, 256, 1, 1, 0, 1), "l1.c2": (64, 63, 64, 3, 1, 1, 1), "l2.0.ds": (256, 63, 512, 3, 2, 0, 1),
          "stem": (3, 255, 64, 7, 2, 0, 1)}
for B in (() if os.environ.get('AB_E2E_ONLY') else (8, 64)):
    for name, (cin, hw, cout, k, st, pad, dil) in SHAPES.items():
        row = []
        for rep in range(2):
            for v in vals:
                _lib.tune(**{knob: v})
                row.append(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, iters=20))
        n = len(vals)
        print("B=%-3d %-8s " % (B, name) + " | ".join("%s=%d: %7.2f %7.2f" % (knob, vals[i], row[i], row[n + i]) for i in range(n)), flush=True)

def e2e(total_b, steps=100):
    dev = torch.device("cuda", 0)
    m = build("sharp", dtype="f16", max_batch=total_b, graph=True)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    m = m.eval().to(dev)
    m.template(torch.from_numpy(synth.image_batch(total_b, 127, stream0=0)).to(dev))
    x = torch.from_numpy(synth.image_batch(total_b, 255, stream0=1000)).to(dev)
    tw = torch.full((total_b, 2), 70.0, device=dev)
    for _ in range(15): m.track_step(x, tw, refine=True, stage=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): m.track_step(x, tw, refine=True, stage=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for B in (8, 64, 1):
    res = []
    for rep in range(2):
        for v in vals:
            _lib.tune(**{knob: v})
            res.append(e2e(B, 100 if B < 64 else 30))
    n = len(vals)
    print("e2e B=%d " % B + " | ".join("%s=%d: %.3f %.3f ms" % (knob, vals[i], res[i], res[n + i]) for i in range(n)), flush=True)
