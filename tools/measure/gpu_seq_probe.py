#!/usr/bin/env python
"""Round 3: what bounds conv_seq_kernel's layers?  Micro-benchmarks through smk_op_conv_seq (per-layer stamps of team 0 / slot 0).

For the layer3 shapes at the bench's batch (B = 8: one 31x31 image per XCD) a sequence of N identical layers, all reading the
same input, one barrier behind each:
  cold   : every layer has its OWN weights (first touch: they come from the Infinity Cache / HBM)
  warm   : every layer uses the SAME weights (L2-resident from the second layer on)
x K-loop stagger off / on x (64x128 only) the deep-ring variant.  The first layer of a sequence is dropped (cold everything)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import ops

SHAPES = [
    # name, cin, cout, k, dil, res, tiles to try
    ("l3.c1", 1024, 256, 1, 1, False, [(64, 128), "deep"]),
    ("l3.c2", 256, 256, 3, 2, False, [(64, 128), "deep"]),
    ("l3.c3", 256, 1024, 1, 1, True, [(128, 256), (64, 256)]),
    ("l2.c1", 512, 128, 1, 1, False, [(64, 64)]),
    ("l2.c2", 128, 128, 3, 1, False, [(64, 64)]),
    ("l2.c3", 128, 512, 1, 1, True, [(64, 256)]),
    ("l3.0.ds", 512, 1024, 3, 1, False, [(128, 256)]),
]
N = 6


def main():
    rng = np.random.default_rng(0)
    print("%-8s %-10s %-5s %-6s | per-layer tiles us (layers 2..%d)            | mean tiles  arrive | launch us" % (
        "layer", "tile", "kstag", "wts", N))
    for name, cin, cout, k, dil, res, tiles in SHAPES:
        hw = 63 if name.startswith("l2.c1") and False else 31
        x = torch.from_numpy(rng.uniform(-1, 1, size=(8, cin, hw, hw)).astype(np.float32)).cuda()
        ws = [(rng.uniform(-1, 1, size=(cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32) for _ in range(N)]
        r = None
        if res:
            r = torch.from_numpy(rng.uniform(-1, 1, size=(8, cout, hw, hw)).astype(np.float32)).cuda()
        for tile in tiles:
            for kstag in (0, 1):
                for wts in ("cold", "warm"):
                    layers = []
                    for i in range(N):
                        l = dict(w=ws[i] if wts == "cold" else ws[0], pad=dil * (k // 2), dil=dil, relu=True, src=-1, tile=tile,
                                 kstag=kstag)
                        layers.append(l)
                    if res:
                        # the residual must be an earlier tensor of the sequence: layer 0 produces it (same shape as the outputs)
                        layers = [dict(w=ws[0], pad=dil * (k // 2), dil=dil, src=-1, tile=tile, kstag=kstag)] + [
                            dict(l, res=0, res_mode=1) for l in layers]
                    _, us, clk = ops.conv_seq(x, layers, iters=6, want_outputs=False)
                    t = clk[2 if res else 1:, 0]
                    a = clk[2 if res else 1:, 1]
                    print("%-8s %-10s %-5d %-6s | %s | %6.2f  %6.2f | %7.1f" % (
                        name, str(tile), kstag, wts, " ".join("%6.2f" % v for v in t), t.mean(), a.mean(), us), flush=True)


if __name__ == "__main__":
    main()
