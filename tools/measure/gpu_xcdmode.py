#!/usr/bin/env python
"""Tile order vs fabric traffic: tm-major (every XCD streams ALL weight panels for each group of M tiles) against tn-major
(every XCD owns a range of weight panels, the activation tiles are read by several XCDs) on the long-K layers; conv_igemm."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401

from siammask_amd import _lib, ops
from gpu_convbench import LAYERS

CASES = [(64, "l3.0.ds", (256, 128), 3), (64, "l3.0.ds", (128, 128), 2), (8, "l3.0.ds", (128, 128), 2), (64, "l3.c2", (128, 128), 2),
         (64, "l3.c1", (128, 128), 2), (64, "l3.c3", (128, 128), 2), (64, "conv_search", (256, 128), 3), (8, "conv_search", (128, 128), 2),
         (64, "l2.0.ds", (256, 128), 3), (8, "l3.c1", (64, 128), 3)]
for B, name, tile, stg in CASES:
    cin, hw, cout, k, st, pad, dil, r, nchw, win, pm, pa, bm = LAYERS[name]
    row = []
    for mode in (1, 2, 0, 1, 2):
        _lib.tune(xcd_mode=mode)
        us = ops.bench_conv(B * bm, cin, hw, hw, cout, k, st, pad, dil, tile=tile, kt=128, stages=stg, dtype="f16", res=bool(r), iters=30)
        row.append("mode%d %.2f" % (mode, us))
    print("B=%-2d %-12s %dx%d : %s" % (B, name, tile[0], tile[1], " | ".join(row)), flush=True)
_lib.tune(xcd_mode=1)
