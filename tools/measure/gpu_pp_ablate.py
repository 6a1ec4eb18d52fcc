"""conv_pp_kernel's K loop with parts removed / re-placed (MEASURE build: SMK_LIB=build_variants/measure/siammask_amd/libsiammask_hip.so,
smk_tune "ablate"): 1 no LDS-DMA, 2 no fragment reads, 4 no MFMAs, 8 no s_setprio, 16 fragment reads retired before the barrier.
us per launch, and cycles per barrier interval of a workgroup at an assumed 2.0 GHz (intervals = 2 * 4 * K tiles * rounds)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
arms = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 8, 16, 24, 1, 2, 3, 4, 5, 6, 7, 0]
cases = [("l3.c2", 256, 31, 256, 3, 1, 2, 2), ("l3.0.ds", 512, 31, 1024, 3, 1, 1, 1)]
names = {0: "full", 1: "no DMA", 2: "no reads", 3: "MFMA + barriers only", 4: "no MFMA", 5: "reads + barriers only", 6: "DMA + barriers only",
         7: "barriers only", 8: "no setprio", 16: "lgkmcnt before barrier", 24: "no setprio + lgkmcnt before barrier", 32: "stage before reads", 67: "MFMA + barriers only, RANDOM operand registers", 66: "no reads, DMA + MFMA, random operand registers"}
for name, cin, hw, cout, k, st, pad, dil in cases:
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // st + 1
    M = B * ho * ho
    fl = 2.0 * M * cout * cin * k * k
    ntile = -(-M // 256) * -(-cout // 256)
    rounds = -(-ntile // 256)
    nk = cin * k * k // 64
    for a in arms:
        _lib.tune(ablate=a)
        us = sorted(ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, pp=True, dtype="f16", iters=iters) for _ in range(3))[1]
        print("%-8s ablate %2d %-36s %8.1f us  %6.0f TF/s-equivalent  %6.0f cycles/interval @2.0 GHz" %
              (name, a, names.get(a, "?"), us, fl / us / 1e6, us * 2000.0 / (rounds * nk * 8)), flush=True)
_lib.tune(ablate=0)
