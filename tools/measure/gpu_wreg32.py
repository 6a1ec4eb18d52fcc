"""32 x 64 conv_wreg tiles for under-filled launches (half the activation rows per workgroup, twice the workgroups): isolated launches"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops, _lib

cases = [("l2.c1 f16 B1", 1, 512, 31, 128, 1, 1, 0, 1), ("l2.c2 f16 B1", 1, 128, 31, 128, 3, 1, 1, 1), ("l2.c3 f16 B1", 1, 128, 31, 512, 1, 1, 0, 1),
         ("l3.c1 f16 B1", 1, 1024, 31, 256, 1, 1, 0, 1), ("l3.c2 f16 B1", 1, 256, 31, 256, 3, 1, 2, 2), ("l3.c3 f16 B1", 1, 256, 31, 1024, 1, 1, 0, 1),
         ("l3.0.ds f16 B1", 1, 512, 31, 1024, 3, 1, 1, 1), ("search f16 B1", 1, 256, 31, 768, 3, 1, 0, 1),
         ("v2.0 f16 B8", 8, 512, 15, 128, 3, 1, 1, 1), ("v1.0 f16 B8", 8, 256, 31, 64, 3, 1, 1, 1), ("l3.c2 f16 B2", 2, 256, 31, 256, 3, 1, 2, 2)]
for name, B, cin, hw, cout, k, st, pad, dil in cases:
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // st + 1
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    row = []
    for tile in ((64, 64), (32, 64)):
        for depth in (3, 4):
            _lib.tune(wreg_stages=depth)
            us = min(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=tile, stages=0, wreg=True, dtype="f16", iters=30) for _ in range(3))
            wgs = ((B * ho * ho + tile[0] - 1) // tile[0]) * ((cout + 63) // 64)
            row.append("%dx%d s%d: %5.1f us (%3d wgs)" % (tile[0], tile[1], depth, us, wgs))
    _lib.tune(wreg_stages=0)
    print("%-16s %s" % (name, " | ".join(row)), flush=True)
