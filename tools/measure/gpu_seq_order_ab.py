#!/usr/bin/env python
"""Round 3 (library built with `make MEASURE=1`): the k-step issue order of wreg_tile, first version vs interleaved,
inside conv_seq_kernel on the bench's layer shapes (B = 8, one 31x31 image per XCD).  Alternating arms, same process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import ops

SHAPES = [("l3.c1", 1024, 256, 1, 1, (64, 128), "old64x128"), ("l3.c2", 256, 256, 3, 2, (64, 128), "old64x128"),
          ("l3.c3", 256, 1024, 1, 1, (128, 256), "old128x256"), ("l2.c1", 512, 128, 1, 1, (64, 64), "old64x64"),
          ("l2.c2", 128, 128, 3, 1, (64, 64), "old64x64"), ("l2.c3", 128, 512, 1, 1, (64, 256), "old64x256"),
          ("l3.0.ds", 512, 1024, 3, 1, (128, 256), "old128x256")]
N = 6
rng = np.random.default_rng(0)
for name, cin, cout, k, dil, new, old in SHAPES:
    x = torch.from_numpy(rng.uniform(-1, 1, size=(8, cin, 31, 31)).astype(np.float32)).cuda()
    ws = [(rng.uniform(-1, 1, size=(cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32) for _ in range(N)]
    res = {}
    for rep in range(3):
        for arm, tile in (("new", new), ("old", old)):
            layers = [dict(w=ws[i], pad=dil * (k // 2), dil=dil, relu=True, src=-1, tile=tile, kstag=1) for i in range(N)]
            _, us, clk = ops.conv_seq(x, layers, iters=6, want_outputs=False)
            res.setdefault(arm, []).append(float(clk[1:, 0].mean()))
    print("%-8s %-10s | tiles us per layer: interleaved %s | first version %s | %+.1f %%" % (
        name, new, " ".join("%6.2f" % v for v in res["new"]), " ".join("%6.2f" % v for v in res["old"]),
        100.0 * (np.mean(res["new"]) / np.mean(res["old"]) - 1)), flush=True)
