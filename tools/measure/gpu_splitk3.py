#!/usr/bin/env python
"""real split-K (with the reduction) per launch on the shapes the probe looked at"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from siammask_amd import _lib, ops
SH = {"v2.0": (512, 15, 128, 3, 1, 1, 1), "l3.c2": (256, 31, 256, 3, 1, 2, 2), "l3.c1": (1024, 31, 256, 1, 1, 0, 1),
      "l2.c2": (128, 31, 128, 3, 1, 1, 1)}
for B in (1, 8):
    for name, (cin, hw, cout, k, st, pad, dil) in SH.items():
        for tile in ((64, 64), (64, 128)):
            row = []
            for sp in (0, 2, 4, 0, 2, 4):
                _lib.tune(ksplit=sp)
                row.append(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, iters=30, tile=tile))
            print("B=%d %-6s tile %-9s off %6.2f %6.2f | x2 %6.2f %6.2f | x4 %6.2f %6.2f us" %
                  (B, name, tile, row[0], row[3], row[1], row[4], row[2], row[5]), flush=True)
_lib.tune(ksplit=1)
