#!/usr/bin/env python
"""Round 3, probe 3 (library built with `make MEASURE=1`): the K-tile time of a 64x128 sequence tile with parts of the
loop removed -- operand refills, MFMAs, A-fragment reads, the K-loop barrier."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import ops

N = 5
WHAT = {(64, 128): "full loop", "no_a": "no A refills", "no_w": "no W refills", "no_mfma": "no MFMA", "abl10": "MFMA + frag reads + barriers",
        "abl11": "frag reads + barriers", "abl12": "barriers only", "abl13": "all but K-loop barriers", "abl14": "all but frag reads"}


def run(name, cin, cout, k, dil, B, hw, tile):
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.uniform(-1, 1, size=(B, cin, hw, hw)).astype(np.float32)).cuda()
    ws = [(rng.uniform(-1, 1, size=(cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32) for _ in range(N)]
    layers = [dict(w=ws[i], pad=dil * (k // 2), dil=dil, relu=True, src=-1, tile=tile, kstag=0) for i in range(N)]
    _, us, clk = ops.conv_seq(x, layers, iters=5, want_outputs=False)
    t = clk[1:, 0]
    nk = cin * k * k // 64
    print("%-6s B=%d %-30s | %s | mean %6.2f us = %.3f us per K tile (%d)" % (
        name, B, WHAT[tile], " ".join("%6.2f" % v for v in t), t.mean(), t.mean() / nk, nk), flush=True)


for name, cin, cout, k, dil in (("l3.c2", 256, 256, 3, 2), ("l3.c1", 1024, 256, 1, 1)):
    for tile in WHAT:
        run(name, cin, cout, k, dil, 8, 31, tile)
