"""B = 64 layer3 conv2 / downsample shapes on the per-launch 128x256 (and 64x128) register-fed tile with parts of the K loop
removed (smk_tune "ablate" bits: 1 no A refills, 2 no W refills, 4 no MFMA, 8 no K-loop barriers, 16 no A-fragment reads,
32 the first version's issue order, 64 nothing removed)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops, _lib

B = 64
WHAT = [(27, "MFMA only"), (11, "MFMA + frag reads"), (3, "MFMA + frag reads + barriers"), (64, "full loop (measurement kernel)"),
        (32, "first version's issue order: full loop"), (0, "production kernel"), (1, "no A refills"), (2, "no W refills"), (3, "no A, no W refills"), (4, "no MFMA"), (7, "frag reads + barriers"),
        (23, "barriers only"), (8, "all but K-loop barriers"), (16, "all but frag reads")]
for name, cin, hw, cout, k, pad, dil in (("l3.c2", 256, 31, 256, 3, 2, 2), ("l3.0.ds", 512, 31, 1024, 3, 2, 2)):
    M = B * hw * hw
    for tile in ((128, 256), (64, 128)):
        ntile = -(-M // tile[0]) * -(-cout // tile[1])
        rounds = -(-ntile // 256)
        nk = cin * k * k // 64
        for ab, what in WHAT:
            _lib.tune(ablate=ab)
            us = ops.bench_conv(B, cin, hw, hw, cout, k, 1, pad, dil, tile=tile, stages=3, wreg=True, dtype="f16", iters=10)
            print("%-8s tile %-10s %-26s %8.1f us  rounds %2d  us/Ktile %.3f" % (name, tile, what, us, rounds, us / rounds / nk), flush=True)
_lib.tune(ablate=0)
