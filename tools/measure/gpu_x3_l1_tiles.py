"""split-operand contexts, layer1 at B = 8 (M = 31 752): conv_wreg tiles against the generic kernel's choice, as plain fp16 convolutions on three times the channels
(the unfused operand: an upper bound for the fused-order packs, which refill a third less)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops

B = 8
cases = [("l1.c1", 768, 63, 64, 1, 1, 0, 1, False), ("l1.c2", 192, 63, 64, 3, 1, 1, 1, False), ("l1.c3", 192, 63, 256, 1, 1, 0, 1, True), ("l1.0.c1", 192, 63, 64, 1, 1, 0, 1, False),
         ("l2.0.c1", 768, 63, 128, 1, 1, 0, 1, False), ("stem", 24, 255, 64, 7, 2, 0, 1, False)]
tiles = [(64, 64), (64, 128), (64, 256), (128, 64), (128, 128), (128, 256), (32, 64)]
for name, cin, hw, cout, k, st, pad, dil, res in cases:
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // st + 1
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    row = []
    for t in tiles:
        if t[1] > max(64, cout):
            continue
        try:
            us = min(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=t, stages=4, wreg=True, dtype="f16", iters=20, res=res) for _ in range(2))
            row.append("%dx%d %6.1f" % (t[0], t[1], us))
        except Exception as e:
            row.append("%dx%d n/a" % t)
    usg = min(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, dtype="f16", iters=20, res=res) for _ in range(2))
    print("%-8s wreg us: %s | igemm auto %6.1f us" % (name, "  ".join(row), usg), flush=True)
