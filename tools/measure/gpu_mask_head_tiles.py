"""the 63x63 mask head (mask_model.mask.head.3: 1x1, 256 -> 3969, NCHW f32 logits; experiments/siammask_sharp/custom.py:89-96) at B = 64 / 8:
conv_igemm_kernel's NCHW epilogue by tile shape -- the run length of its stores is the tile's row count x 4 bytes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops
for B in (64, 8):
    mb = B * 3969 * 625 * 4 / 1e6
    for tile in ((128, 128), (256, 128), (128, 64), (64, 128)):
        for st in (0,):
            us = min(ops.bench_conv(B, 256, 25, 25, 3969, 1, nchw=True, tile=tile, dtype="f16", iters=20) for _ in range(3))
            print("B=%d tile %s: %7.1f us  %6.0f GB/s of logits  %5.0f TF/s" % (B, tile, us, mb / us * 1e3 / 1e3, 2.0 * B * 625 * 3969 * 256 / us / 1e6), flush=True)
