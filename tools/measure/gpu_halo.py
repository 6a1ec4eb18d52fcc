#!/usr/bin/env python
"""conv3x3_halo_kernel against the generic implicit-GEMM kernel: per-shape launch time, then the whole step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from siammask_amd import _lib, ops, synth
from siammask_amd.custom import build
SHAPES = {"l1.c2": (64, 63, 64, 1, 1), "l2.c2": (128, 31, 128, 1, 1), "l3.0.c2": (256, 31, 256, 1, 1),
          "l3.c2": (256, 31, 256, 2, 2), "l3.0.ds": (512, 31, 1024, 1, 1), "conv_search": (256, 31, 256, 0, 1),
          "rpn.head3": (256, 25, 256, 1, 1)}
for B in (8, 64, 1):
    for name, (cin, hw, cout, pad, dil) in SHAPES.items():
        row = []
        for rep in range(2):
            row.append(ops.bench_conv(B, cin, hw, hw, cout, 3, 1, pad, dil, iters=20))
            for t in ((128, 128), (64, 128)):
                try:
                    row.append(ops.bench_conv(B, cin, hw, hw, cout, 3, 1, pad, dil, iters=20, halo=True, tile=t))
                except RuntimeError:
                    row.append(float("nan"))
        print("B=%-3d %-12s generic %7.2f %7.2f | halo128 %7.2f %7.2f | halo64 %7.2f %7.2f us" %
              (B, name, row[0], row[3], row[1], row[4], row[2], row[5]), flush=True)

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
vals = (0, 1)
for B in (8, 64, 1):
    res = []
    for rep in range(2):
        for v in vals:
            _lib.tune(halo=v)
            res.append(e2e(B, 100 if B < 64 else 30))
    n = len(vals)
    print("e2e B=%d " % B + " | ".join("halo=%d: %.3f %.3f ms" % (vals[i], res[i], res[n + i]) for i in range(n)), flush=True)
