#!/usr/bin/env python
"""A/B of the consumer-wave priority (smk_tune prio) on the heavy conv shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from siammask_amd import _lib, ops
SHAPES = {"l3.0.ds": (512, 31, 1024, 3, 1, 1, 1), "l3.c2": (256, 31, 256, 3, 1, 2, 2), "l3.c3": (256, 31, 1024, 1, 1, 0, 1),
          "l3.c1": (1024, 31, 256, 1, 1, 0, 1)}
for B in (8, 64):
    for name, (cin, hw, cout, k, st, pad, dil) in SHAPES.items():
        for tile, kt, stages in (((128, 128), 128, 2), ((128, 128), 128, 3), ((256, 128), 128, 3), ((64, 128), 128, 3), ((64, 64), 256, 2)):
            row = []
            for rep in range(2):
                for prio in (0, 1, 2, 3, -1):
                    _lib.tune(prio=prio)
                    row.append(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=tile, kt=kt, stages=stages, iters=20))
            _lib.tune(prio=0)
            print("B=%-3d %-8s %dx%d/%d s%d  prio0 %7.2f %7.2f | p1 %7.2f %7.2f | p2 %7.2f %7.2f | p3 %7.2f %7.2f | prod1 %7.2f %7.2f" % (
                B, name, tile[0], tile[1], kt, stages, row[0], row[5], row[1], row[6], row[2], row[7], row[3], row[8], row[4], row[9]), flush=True)
