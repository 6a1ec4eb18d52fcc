#!/usr/bin/env python
"""What sets the K-tile time of conv_wreg_kernel's 64x64 tile (K split over the four consumer waves)?  Needs a MEASURE=1 library with the
<2,1,4> ablation kernels (SMK_LIB; round 4: build_variants/measure_src).  smk_tune "ablate" bits: 1 no A refills, 2 no W refills,
4 no MFMA, 8 no K-loop barriers, 16 no A-fragment reads.  Warm operands (30 launches back to back), us per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: F401

from siammask_amd import _lib, ops
from gpu_convbench import LAYERS

assert _lib.tune_get("measure_build") == 1, "needs a MEASURE=1 library (SMK_LIB)"
CASES = [(8, "v2.0"), (8, "v1.0"), (1, "l3.c1"), (1, "l3.c3"), (1, "l3.c2"), (1, "l3.0.ds"), (1, "l2.c1"), (8, "l3.c1")]
for B, name in CASES:
    cin, hw, cout, k, st, pad, dil, r, nchw, win, pm, pa, bm = LAYERS[name]
    row = []
    for ab, tag in ((0, "full"), (1, "noA"), (2, "noW"), (3, "noA+noW"), (4, "noMFMA"), (8, "noBar"), (16, "noFrag"), (11, "noA+noW+noBar"), (27, "noA+noW+noBar+noFrag"), (23, "onlyBar"), (0, "full")):
        _lib.tune(ablate=ab)
        us = ops.bench_conv(B * bm, cin, hw, hw, cout, k, st, pad, dil, tile=(64, 64), stages=3, wreg=True, dtype="f16",
                            res=bool(r), win=win, pos_mul=pm, pos_add=pa, iters=30)
        row.append("%s %.2f" % (tag, us))
    Hl = win[0] if win else hw
    Ho = (Hl + 2 * pad - dil * (k - 1) - 1) // st + 1
    nk = (k * k * cin + 63) // 64
    print("B=%-2d %-8s M=%d N=%d K=%d (%d K tiles) : %s" % (B, name, B * bm * Ho * Ho, cout, k * k * cin, nk, " | ".join(row)), flush=True)
_lib.tune(ablate=0)
