"""conv_wreg's ACTIVATION ring 3 .. 7 K tiles deep (weights two K tiles ahead as always; `make MEASURE=1` build, smk_tune wreg_stages): the ablation
(tools/measure/gpu_wreg_narrow_ablate.py) says the narrow tiles' K loops wait for the activation refills, 16 KB in flight per CU at ~20 GB/s."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops, _lib

cases = [("l3.c1 x3 B8", 8, 3072, 31, 256, 1, 1, 0, 1, (64, 128)), ("l3.c2 x3 B8", 8, 768, 31, 256, 3, 1, 2, 2, (64, 128)),
         ("l3.c3 x3 B8", 8, 768, 31, 1024, 1, 1, 0, 1, (128, 256)), ("l2.c2 x3 B8", 8, 384, 31, 128, 3, 1, 1, 1, (64, 64)),
         ("l3.c1 f16 B1", 1, 1024, 31, 256, 1, 1, 0, 1, (64, 64)), ("l3.c2 f16 B1", 1, 256, 31, 256, 3, 1, 2, 2, (64, 64)),
         ("l3.c3 f16 B1", 1, 256, 31, 1024, 1, 1, 0, 1, (64, 64)), ("v2.0 f16 B8", 8, 512, 15, 128, 3, 1, 1, 1, (64, 64)),
         ("search f16 B8", 8, 256, 31, 768, 3, 1, 0, 1, (128, 256))]
for name, B, cin, hw, cout, k, st, pad, dil, tile in cases:
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // st + 1
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    row = []
    for depth in (3, 4, 5, 6, 7):
        _lib.tune(wreg_stages=depth)
        us = min(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=tile, stages=0, wreg=True, dtype="f16", iters=20) for _ in range(3))
        row.append("%d: %.1f us (%.0f TF)" % (depth, us, fl / us / 1e6))
    _lib.tune(wreg_stages=0)
    print("%-14s %dx%d  %s" % (name, tile[0], tile[1], " | ".join(row)), flush=True)
