#!/usr/bin/env python
"""Round 3, probe 2: is the K loop of a sequence layer bound per CU or by something the CUs share?

Same layer (layer3 conv1 / conv2 shapes, 64x128 tiles), N identical layers per launch, per-layer time of (team 0, slot 0):
  * 32 workgroups of the XCD busy (31x31 image: 16 x 2 tiles) vs 16 busy (22x22: 8 x 2 tiles)  -> shared inside the XCD?
  * 8 XCDs busy (B = 8) vs one (B = 1)                                                          -> shared across the chip?
  * ablations of the tile routine: no activation refills / no weight refills / no MFMA         -> which stream sets the time?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import ops

N = 5


def run(name, cin, cout, k, dil, B, hw, tile, kstag):
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(B, cin, hw, hw)).astype(np.float32)).cuda()
    ws = [(rng.uniform(-1, 1, size=(cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32) for _ in range(N)]
    layers = [dict(w=ws[i], pad=dil * (k // 2), dil=dil, relu=True, src=-1, tile=tile, kstag=kstag) for i in range(N)]
    _, us, clk = ops.conv_seq(x, layers, iters=5, want_outputs=False)
    t = clk[1:, 0]
    nk = cin * k * k // 64
    print("%-6s B=%d %2dx%-2d tiles/XCD %2d  %-8s kstag %d | %s | mean %6.2f us = %.3f us per K tile (%d)" % (
        name, B, hw, hw, ((hw * hw + 63) // 64) * ((cout + 127) // 128), str(tile), kstag, " ".join("%6.2f" % v for v in t),
        t.mean(), t.mean() / nk, nk), flush=True)


for name, cin, cout, k, dil in (("l3.c1", 1024, 256, 1, 1), ("l3.c2", 256, 256, 3, 2)):
    for B, hw in ((8, 31), (8, 22), (8, 15), (1, 31), (1, 22)):
        for kstag in (0, 1):
            run(name, cin, cout, k, dil, B, hw, (64, 128), kstag)
    for tile in ("no_a", "no_w", "no_mfma"):
        run(name, cin, cout, k, dil, 8, 31, tile, 0)
        run(name, cin, cout, k, dil, 1, 31, tile, 0)
