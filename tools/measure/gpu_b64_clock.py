"""Effective shader clock of the B = 64 128x256 K loop with parts removed: run under
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace; tools/measure/clock_stats.py divides the counter by the dispatch duration."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops, _lib

for ab in (27, 11, 3, 7, 4, 1, 2, 64, 32, 0):
    _lib.tune(ablate=ab)
    us = ops.bench_conv(64, 512, 31, 31, 1024, 3, 1, 2, 2, tile=(128, 256), stages=3, wreg=True, dtype="f16", iters=10)
    print(ab, us, flush=True)
_lib.tune(ablate=0)
