#!/usr/bin/env python
"""Sharp fp16 fused step (graph replay) with the persistent per-XCD sequence forced on for EVERY batch size vs off, alternating
in one process: for which B does conv_seq_kernel pay?  (The product enables it for seq_min_batch <= B <= seq_max_batch and for the multiples of 8 up to seq_mult_max.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_seq_ab import run

for B in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,5,6,7,8,10,12,16,24,32").split(",")]:
    ex = {"seq_min_batch": 1, "seq_max_batch": 64, "seq_mult_max": 64}
    d0, _, o0 = run(B, 0, extra=ex)
    d1, s1, o1 = run(B, 1, extra=ex)
    d0b, _, _ = run(B, 0, extra=ex)
    d1b, _, _ = run(B, 1, extra=ex)
    err = max(float((o0[k].double() - o1[k].double()).abs().max() / (o0[k].double().abs().max() + 1e-30)) for k in o0)
    print("B=%-3d seq off %.4f / %.4f ms   seq on %.4f / %.4f ms   x%.3f   frames/s %7.0f -> %7.0f   status %s   max rel diff %.1e" % (
        B, d0, d0b, d1, d1b, min(d0, d0b) / min(d1, d1b), B / min(d0, d0b) * 1e3, B / min(d1, d1b) * 1e3, s1, err), flush=True)
