#!/usr/bin/env python
"""A/B in one process (sharp fp16 fused step, graph replay): tn-major order for the NCHW mask head, and the fused
cls3 + loc3 + decode launch; plus the per-launch profile of the tail with everything on."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_seq_ab import run

for B in (8, 64, 1):
    row = []
    for tag, kw in (("base", dict(nchw_tn_major=0, heads_decode=0)), ("tn", dict(nchw_tn_major=1, heads_decode=0)),
                    ("hd", dict(nchw_tn_major=0, heads_decode=1)), ("both", dict(nchw_tn_major=1, heads_decode=1)),
                    ("base", dict(nchw_tn_major=0, heads_decode=0)), ("both", dict(nchw_tn_major=1, heads_decode=1))):
        d, st, _ = run(B, 1, steps=60 if B == 64 else 150, extra=kw)
        row.append("%s %.4f" % (tag, d))
    print("B=%d ms/step: %s" % (B, " | ".join(row)), flush=True)
