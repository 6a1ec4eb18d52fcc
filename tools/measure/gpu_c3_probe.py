#!/usr/bin/env python
"""What does the residual cost the short-K wide-N 1x1 convolutions (bottleneck conv3)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from siammask_amd import ops
SH = {"l1.c3": (64, 63, 256), "l2.c3": (128, 31, 512), "l3.c3": (256, 31, 1024)}
for B in (8, 64):
    for name, (cin, hw, cout) in SH.items():
        for tile in (None, (128, 128), (64, 128), (256, 128)):
            r = [ops.bench_conv(B, cin, hw, hw, cout, 1, 1, 0, 1, iters=30, res=res, tile=tile) for res in (False, True, False, True)]
            print("B=%-2d %-6s tile %-10s no-res %6.2f %6.2f | res %6.2f %6.2f us" % (B, name, tile, r[0], r[2], r[1], r[3]), flush=True)
