"""B = 64 layer3 convs alone (conv_wreg 128x256): us per launch, us per K tile of a workgroup.  Run under rocprofv3 --pmc for the
LDS / wait counters of exactly these launches."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cases = [("l3.c2", 256, 31, 256, 3, 1, 2, 2), ("l3.0.ds", 512, 31, 1024, 3, 1, 2, 2), ("conv_search", 256, 31, 256, 3, 1, 0, 1),
         ("l3.c1", 1024, 31, 256, 1, 1, 0, 1), ("l3.c3", 256, 31, 1024, 1, 1, 0, 1)]
for name, cin, hw, cout, k, st, pad, dil in cases:
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // st + 1
    M = B * ho * ho
    for tile in ((128, 256), (64, 256), (128, 128)):
        us = ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=tile, stages=3, wreg=True, dtype="f16", iters=iters,
                            res=(name == "l3.c3"))
        ntile = -(-M // tile[0]) * -(-cout // tile[1])
        rounds = -(-ntile // 256)
        nk = cin * k * k // 64
        fl = 2.0 * M * cout * cin * k * k
        print("%-12s tile %-10s %8.1f us  %7.0f TF/s  tiles %5d rounds %2d  us/Ktile %.3f" %
              (name, tile, us, fl / us / 1e6, ntile, rounds, us / rounds / nk), flush=True)
