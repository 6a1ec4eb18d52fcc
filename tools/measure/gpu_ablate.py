#!/usr/bin/env python
"""Which stream sets conv_wreg_kernel's K-tile time?  Compile-time ablation builds (smk_tune "ablate": 1 = the activation
tiles are staged once and never refilled, 2 = the weight fragments are loaded once and never refilled, 4 = no MFMAs; results
are wrong by construction, timing only), per layer geometry, us per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401

from siammask_amd import _lib, ops
from gpu_convbench import LAYERS

CASES = [(8, "l3.c1", (64, 128)), (8, "l3.c2", (64, 128)), (8, "l3.c3", (64, 256)), (8, "l3.0.ds", (128, 256)),
         (8, "l3.0.ds", (64, 256)), (64, "l3.c1", (128, 256)), (64, "l3.0.ds", (128, 256))]
for B, name, tile in CASES:
    cin, hw, cout, k, st, pad, dil, r, nchw, win, pm, pa, bm = LAYERS[name]
    row = []
    for ab, tag in ((0, "full"), (1, "noA"), (2, "noW"), (3, "noA+noW"), (4, "noMFMA"), (7, "nothing"), (0, "full")):
        _lib.tune(ablate=ab)
        us = ops.bench_conv(B * bm, cin, hw, hw, cout, k, st, pad, dil, tile=tile, stages=3, wreg=True, dtype="f16",
                            res=bool(r), iters=30)
        row.append("%s %.2f" % (tag, us))
    print("B=%-2d %-8s %dx%d : %s" % (B, name, tile[0], tile[1], " | ".join(row)), flush=True)
_lib.tune(ablate=0)
