"""one kernel, many launches, for rocprofv3 --pmc: argv = pp|wreg layer-name B iters"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from siammask_amd import ops
which, layer, B, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
cases = {"l3.c2": (256, 31, 256, 3, 1, 2, 2), "l3.0.ds": (512, 31, 1024, 3, 1, 1, 1), "conv_search": (256, 31, 768, 3, 1, 0, 1),
         "l2.0.ds": (256, 63, 512, 3, 2, 0, 1)}
cin, hw, cout, k, st, pad, dil = cases[layer]
kw = dict(pp=True) if which == "pp" else dict(tile=(128, 256), stages=3, wreg=True)
print(which, layer, ops.bench_conv(B, cin, hw, hw, cout, k, st, pad, dil, dtype="f16", iters=iters, **kw))
