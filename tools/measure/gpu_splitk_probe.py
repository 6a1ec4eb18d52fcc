#!/usr/bin/env python
"""Would splitting K across workgroups help the under-filled launches?  Emulation without the reduction:
time(B*s streams, Cin/s channels) has the same MACs, s x the workgroups and 1/s of the K loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from siammask_amd import ops
SH = {"l3.c2": (256, 31, 256, 3, 1, 2, 2), "l3.c1": (1024, 31, 256, 1, 1, 0, 1), "l3.c3": (256, 31, 1024, 1, 1, 0, 1),
      "l2.c2": (128, 31, 128, 3, 1, 1, 1), "l2.c1": (512, 31, 128, 1, 1, 0, 1), "v2.0": (512, 15, 128, 3, 1, 1, 1),
      "l3.0.ds": (512, 31, 1024, 3, 1, 1, 1)}
for B in (1, 8):
    for name, (cin, hw, cout, k, st, pad, dil) in SH.items():
        row = []
        for s in (1, 2, 4):
            halo = k == 3 and (cin // s) % 64 == 0 and name != "l3.0.ds"
            best = 1e9
            for tile in ((None,) if not halo else (None, (64, 128))):
                for hl in ((False, True) if halo and tile else (False,)):
                    try:
                        best = min(best, ops.bench_conv(B * s, cin // s, hw, hw, cout, k, st, pad, dil, iters=30, halo=hl, tile=tile))
                    except RuntimeError:
                        pass
            row.append(best)
        print("B=%d %-8s split1 %6.2f  split2 %6.2f  split4 %6.2f us" % (B, name, *row), flush=True)
