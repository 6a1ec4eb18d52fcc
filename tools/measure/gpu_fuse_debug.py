#!/usr/bin/env python
"""Round 3: the fused (conv3, next 1x1) tile routine of conv_seq_kernel against the unfused list on a small case -- where do
they differ?  Prints, per pair shape, the max difference of conv3's output and of the 1x1's output, and if they disagree the
rows (mod 32) and the 32-channel blocks that are off (a layout mistake shows as a pattern)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import _lib, ops

for cin, planes in ((512, 128), (1024, 256)):
    rng = np.random.default_rng(cin)
    B, S = 2, 9                                   # 81 rows per image: 2.5 tiles of 32
    x = rng.uniform(-1, 1, size=(B, cin, S, S)).astype(np.float32)

    def w(co, ci, k):
        return (rng.uniform(-1, 1, size=(co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32)

    layers = [dict(w=w(planes, cin, 1), relu=True), dict(w=w(planes, planes, 3), pad=1, relu=True),
              dict(w=w(cin, planes, 1), b=rng.uniform(-1, 1, cin).astype(np.float32), relu=True, res=-1, res_mode=1),
              dict(w=w(planes, cin, 1), b=rng.uniform(-1, 1, planes).astype(np.float32), relu=True)]
    xd = torch.from_numpy(x).cuda()
    info = {}
    _lib.tune(seq_fuse=0)
    plain, _, _ = ops.conv_seq(xd, layers, info=info)
    _lib.tune(seq_fuse=1)
    try:
        fused, _, _ = ops.conv_seq(xd, layers, info=info)
    except Exception as e:                        # noqa: BLE001
        print("shape %s: fused launch failed: %s" % ((cin, planes), e))
        continue
    print("shape %s: fused pairs %d" % ((cin, planes), info["fused_pairs"]))
    for i, name in ((2, "conv3"), (3, "1x1")):
        a = fused[i].cpu().numpy().astype(np.float64)
        b = plain[i].cpu().numpy().astype(np.float64)
        d = np.abs(a - b)                         # [B, C, H, W]
        print("  %-5s max|diff| %.3e of max|ref| %.3e  (nan: %d)" % (name, np.nanmax(d), np.abs(b).max(), int(np.isnan(a).sum())))
        if not np.nanmax(d) <= 2e-2 * np.abs(b).max() or np.isnan(a).any():
            d = np.nan_to_num(d, nan=1e9).reshape(B, d.shape[1], S * S)
            rows = d.max(axis=1)                  # [B, pixel]
            bad_rows = sorted({int(p % 32) for bb in range(B) for p in np.nonzero(rows[bb] > 1e-2)[0]})
            blocks = d.reshape(B, d.shape[1] // 32, 32, S * S).max(axis=(0, 2, 3))
            quads = d.reshape(B, d.shape[1] // 4, 4, S * S).max(axis=(0, 2, 3)).reshape(-1, 8).max(axis=0)
            print("    rows (mod 32) off: %s" % bad_rows)
            print("    32-channel blocks off: %s" % np.nonzero(blocks > 1e-2)[0].tolist())
            print("    channel quads (mod 8 quads = 32 ch) off: %s" % np.nonzero(quads > 1e-2)[0].tolist())
            print("    pixels off per image: %s" % [int((rows[bb] > 1e-2).sum()) for bb in range(B)])
