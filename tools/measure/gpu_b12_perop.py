#!/usr/bin/env python
"""Round 3, open item (profiles/r03h_b12_race.txt): is the B = 12 race reproducible through smk_op_conv_seq alone?  layer2's shape as
a caller-described list -- conv1, conv2 (64-pixel patch-sharing tiles), [conv3 + next conv1], conv2, [conv3 + next conv1], conv2,
conv3 -- at B = 12 on 31 x 31 with the pair fusion forced (seq_fuse 3), against the same list unfused, per layer and per image,
several launches: which tensor goes wrong first?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
S, cin, planes = 31, 512, 128
rng = np.random.default_rng(12)


def w(co, ci, k):
    return (rng.uniform(-1, 1, size=(co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32)


def bias(n):
    return rng.uniform(-0.5, 0.5, n).astype(np.float32)


x = torch.from_numpy(rng.uniform(-1, 1, size=(B, cin, S, S)).astype(np.float32)).cuda()
layers, names = [], []
for b in range(3):
    res = len(layers) - 1          # -1 = x for the first block, else the previous block's conv3
    layers += [dict(w=w(planes, cin, 1), b=bias(planes), relu=True), dict(w=w(planes, planes, 3), b=bias(planes), pad=1, relu=True),
               dict(w=w(cin, planes, 1), b=bias(cin), relu=True, res=res, res_mode=1)]
    names += ["b%d.c1" % b, "b%d.c2" % b, "b%d.c3" % b]
info = {}
_lib.tune(seq_fuse=0)
ref, _, _ = ops.conv_seq(x, layers, info=info)
ref = [r.cpu().numpy().astype(np.float64) for r in ref]
print("unfused list: fused pairs %d" % info["fused_pairs"])
for knobs in (dict(seq_fuse=3), dict(seq_fuse=3, seq_kstag_mask=0), dict(seq_fuse=3, seq_halo=0), dict(seq_fuse=3)):
    _lib.tune(seq_kstag_mask=7, seq_halo=1)
    _lib.tune(**knobs)
    for rep in range(4):
        outs, _, _ = ops.conv_seq(x, layers, info=info)
        bad = []
        for i, o in enumerate(outs):
            o = o.cpu().numpy().astype(np.float64)
            per = [float(np.abs(o[b] - ref[i][b]).max() / (np.abs(ref[i][b]).max() + 1e-30)) for b in range(B)]
            if max(per) > 1e-2:
                e = np.abs(o - ref[i]).max(axis=1).reshape(B, -1)            # [B, pixels]
                bb = int(np.argmax(per))
                blocks = sorted(set((np.nonzero(e[bb] > 1e-2 * np.abs(ref[i][bb]).max())[0] // 32).tolist()))
                bad.append("%s: images %s (worst %.1e), 32-pixel blocks of image %d: %s" % (
                    names[i], [b for b in range(B) if per[b] > 1e-2], max(per), bb, blocks))
        print("%s launch %d (fused pairs %d): %s" % (knobs, rep, info["fused_pairs"], "; ".join(bad) if bad else "all layers within 1e-2"), flush=True)
_lib.tune(seq_fuse=1, seq_kstag_mask=7, seq_halo=1)
