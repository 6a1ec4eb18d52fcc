#!/usr/bin/env python
"""Round 3: what bounds the K loop of the patch-sharing tile (wreg_halo_tile.inc)?  The real layer3 identity Bottleneck at the
bench's batch (B = 8, 31 x 31, 1024 -> 256 -> 256 d2 -> 1024 + residual) through smk_op_conv_seq, per-layer stamps of team 0 /
slot 0; run once per library build (weight ring of 3 / 6 k-steps: gpu_halo_probe.sh) and with the im2col tile (seq_halo 0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import _lib, ops

rng = np.random.default_rng(3)


def w(co, ci, k):
    return (rng.uniform(-1, 1, size=(co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32)


x = torch.from_numpy(rng.uniform(-1, 1, size=(8, 1024, 31, 31)).astype(np.float32)).cuda()
layers = []
for b in range(3):
    blk = [dict(w=w(256, 1024, 1), relu=True), dict(w=w(256, 256, 3), pad=2, dil=2, relu=True),
           dict(w=w(1024, 256, 1), relu=True, res=len(layers) - 1, res_mode=1)]
    layers += blk
for halo in (1, 0):
    _lib.tune(seq_halo=halo)
    best = None
    for rep in range(3):
        _, us, clk = ops.conv_seq(x, layers, iters=20, want_outputs=False)
        best = clk[:, 0] if best is None else np.minimum(best, clk[:, 0])
    print("%s seq_halo=%d: %.1f us per launch; conv2 tiles %s us; all layers %s" % (
        sys.argv[1] if len(sys.argv) > 1 else "", halo, us, np.round(best[[1, 4, 7]], 2).tolist(), np.round(best, 1).tolist()), flush=True)
