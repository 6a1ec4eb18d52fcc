#!/usr/bin/env python
"""A/B of one smk_tune knob on the sharp fp16 fused step (graph replay), alternating off/on in one process.
    python tools/measure/gpu_knob_ab.py chain_mask 8,1,64 [v0,v1]        (values default to 0,1; e.g. npw 8 2,4)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_seq_ab import run

knob = sys.argv[1]
v0, v1 = [int(a) for a in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
for B in [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "8").split(",")]:
    row, outs = [], {}
    for v in (v0, v1, v0, v1):
        d, st, o = run(B, 1, steps=60 if B == 64 else 150, extra={knob: v})
        row.append("%s=%d %.4f" % (knob, v, d))
        outs[v] = o
    err = {k: float((outs[v0][k].double() - outs[v1][k].double()).abs().max()) for k in outs[v0]}
    print("B=%d ms/step: %s   max|diff| %s" % (B, " | ".join(row), err), flush=True)
