"""Where does a narrow conv_wreg tile (64 x 128: FM = 2, WN = 2, WK = 2 -- 8 MFMAs per wave between two K-tile barriers) spend its K loop?
The measurement kernels of conv_wreg.hip (`make MEASURE=1`, smk_tune "ablate": 1 no A refills, 2 no W refills, 4 no MFMA, 8 no K-loop barriers,
16 no A-fragment reads; results wrong by construction) on the split-operand contexts' layer3 shapes at B = 8 (operand channels = 3 x stored)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cases = [("l3.c1 x3", 3072, 31, 256, 1, 1, 0, 1), ("l3.c2 x3", 768, 31, 256, 3, 1, 2, 2), ("l3.c3 x3", 768, 31, 1024, 1, 1, 0, 1),
         ("l3.c2 f16", 256, 31, 256, 3, 1, 2, 2)]
arms = [(0, "full"), (64, "ablate build, nothing removed"), (8, "no K-loop barriers"), (1, "no A refills"), (2, "no W refills"), (3, "no refills"),
        (16, "no A-fragment reads"), (4, "no MFMA"), (11, "no refills, no barriers"), (27, "MFMA only")]
for name, cin, hw, cout, k, st, pad, dil in cases:
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // st + 1
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    for tile in ((64, 128), (128, 256)):
        row = []
        for a, what in arms:
            _lib.tune(ablate=a)
            us = min(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=tile, stages=3, wreg=True, dtype="f16", iters=20) for _ in range(2))
            row.append("%s %.1f us (%.0f TF)" % (what, us, fl / us / 1e6))
        _lib.tune(ablate=0)
        print("%-10s %dx%d: %s" % (name, tile[0], tile[1], " | ".join(row)), flush=True)
