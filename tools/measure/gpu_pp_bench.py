"""conv_pp_kernel (256 x 256 tiles, ping-pong wave groups) against conv_wreg_kernel's 128 x 256 tile on the long-K layers of the
B = 64 regime: microseconds per launch (HIP events over back-to-back launches, uniform random operands) and TFLOP/s, interleaved rounds."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cases = [("l3.c2", 256, 31, 256, 3, 1, 2, 2), ("l3.0.ds", 512, 31, 1024, 3, 1, 1, 1), ("conv_search", 256, 31, 768, 3, 1, 0, 1),
         ("l2.0.ds", 256, 63, 512, 3, 2, 0, 1), ("l3.c1", 1024, 31, 256, 1, 1, 0, 1), ("l3.c3", 256, 31, 1024, 1, 1, 0, 1)]
for name, cin, hw, cout, k, st, pad, dil in cases:
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // st + 1
    M = B * ho * ho
    fl = 2.0 * M * cout * cin * k * k
    res = {"pp": [], "wreg": []}
    for r in range(rounds):
        res["wreg"].append(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, tile=(128, 256), stages=3, wreg=True, dtype="f16", iters=iters))
        res["pp"].append(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, pp=True, dtype="f16", iters=iters))
    for kname in ("wreg", "pp"):
        us = sorted(res[kname])[len(res[kname]) // 2]
        bm = 256 if kname == "pp" else 128
        ntile = -(-M // bm) * -(-cout // 256)
        print("%-12s %-5s %8.1f us (min %7.1f)  %7.0f TF/s  tiles %5d = %.2f rounds" %
              (name, kname, us, min(res[kname]), fl / us / 1e6, ntile, ntile / 256.0), flush=True)
