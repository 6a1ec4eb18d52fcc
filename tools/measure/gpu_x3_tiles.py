"""split-operand contexts: which conv_wreg tile for the long-K shapes of layer2 / layer3 at B = 8?  A split-operand convolution IS an fp16 convolution
with three times the input channels, so smk_bench_conv on (3 Cin) channels times exactly its K loop (the splitting epilogue stores three planes
instead of one: not in this number)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cases = [("l3.c1", 3072, 31, 256, 1, 1, 0, 1), ("l3.c2", 768, 31, 256, 3, 1, 2, 2), ("l3.c3", 768, 31, 1024, 1, 1, 0, 1),
         ("l2.c2", 384, 31, 128, 3, 1, 1, 1), ("l2.c1", 1536, 31, 128, 1, 1, 0, 1), ("conv_search", 768, 31, 768, 3, 1, 0, 1),
         ("l3.0.ds", 1536, 31, 1024, 3, 1, 1, 1), ("l2.0.ds", 768, 63, 512, 3, 2, 0, 1)]
tiles = [(64, 64), (64, 128), (64, 256), (128, 128), (128, 256), (128, 64)]
for name, cin, hw, cout, k, st, pad, dil in cases:
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // st + 1
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    row = []
    for t in tiles:
        if t[1] > max(64, cout):
            continue
        us = min(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=t, stages=3, wreg=True, dtype="f16", iters=20, res=(name == "l3.c3")) for _ in range(2))
        row.append("%dx%d %6.1f us (%4.0f TF)" % (t[0], t[1], us, fl / us / 1e6))
    usg = min(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, dtype="f16", iters=20, res=(name == "l3.c3")) for _ in range(2))
    print("%-12s %s | igemm auto %6.1f us" % (name, "  ".join(row), usg), flush=True)
