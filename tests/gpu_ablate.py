#!/usr/bin/env python
"""Ablation of the 128x128x128 conv kernel (measurement only): which of MFMA / LDS-DMA / LDS reads
bounds the main loop.  smk_tune("ablate", n): 0 full, 1 no MFMA, 2 no LDS-DMA, 3 no fragment reads,
4 LDS-DMA only, 5 no LDS-DMA of the weight operand."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from siammask_amd import _lib, ops  # noqa: E402

SHAPES = {"l3.0.ds": (512, 31, 1024, 3, 1, 1, 1), "l3.c2": (256, 31, 256, 3, 1, 2, 2), "l3.c3": (256, 31, 1024, 1, 1, 0, 1)}
out = {}
for B in (8, 64):
    for name, (cin, hw, cout, k, st, pad, dil) in SHAPES.items():
        for tile, stages in (((128, 128), 2), ((128, 128), 3), ((256, 128), 3)):
            row = {}
            for abl in range(5):
                _lib.tune(ablate=abl)
                row[abl] = round(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=tile, kt=128,
                                                stages=stages, iters=20), 2)
            _lib.tune(ablate=0)
            out["B%d %s %dx%d s%d" % (B, name, tile[0], tile[1], stages)] = row
            print("B=%-3d %-8s %dx%d s%d  full %8.2f | noMFMA %8.2f | noDMA %8.2f | noDSread %8.2f | DMAonly %8.2f" % (
                B, name, tile[0], tile[1], stages, row[0], row[1], row[2], row[3], row[4]), flush=True)
json.dump(out, open("gpurun_out/ablate.json", "w"), indent=1)
