#!/usr/bin/env python
"""A/B micro-benchmark: conv_wreg_kernel (weights global -> VGPR) against the best LDS-staged instantiation
(conv_igemm_kernel tiles, conv3x3_halo_kernel) on every heavy convolution geometry of the path.
    python tools/measure/gpu_wregbench.py 8,64 gpurun_out/wregbench.json
Output per batch/layer: microseconds + TFLOP/s of every candidate."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: F401

from siammask_amd import ops
from gpu_convbench import LAYERS

ONLY = ("stem", "l1.0.ds", "l1.c1", "l1.c2", "l1.c3", "l2.0.ds", "l2.0.c1", "l2.0.c2", "l2.c1", "l2.c2", "l2.c3", "l3.0.ds",
        "l3.0.c1", "l3.c1", "l3.c2", "l3.c3", "conv_search", "head0", "v2.0", "v1.0")
BASE = [((128, 128), 128, 2), ((64, 128), 128, 3), ((64, 64), 256, 3), ((128, 64), 128, 3), ((256, 128), 128, 3)]
WREG = [(64, 256), (64, 128), (64, 64), (128, 256), (128, 128), (128, 64)]


def main():
    batches = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8]
    out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/wregbench.json"
    res = {}
    t0 = time.time()
    for B in batches:
        res[B] = {}
        for name in ONLY:
            cin, hw, cout, k, st, pad, dil, r, nchw, win, pm, pa, bm = LAYERS[name]
            Hl = win[0] if win else hw
            Ho = (Hl + 2 * pad - dil * (k - 1) - 1) // st + 1
            gflop = 2.0 * B * bm * Ho * Ho * cout * cin * k * k / 1e9
            kw = dict(dtype="f16", res=bool(r), win=win, pos_mul=pm, pos_add=pa, iters=30)
            runs = {}
            for tile, kt, stg in BASE:
                try:
                    runs["igemm %dx%d" % tile] = ops.bench_conv(B * bm, cin, hw, hw, cout, k, st, pad, dil, tile=tile, kt=kt,
                                                               stages=stg, **kw)
                except Exception as e:  # noqa: BLE001
                    runs["igemm %dx%d" % tile] = "ERR %s" % str(e)[:60]
            if k == 3 and st == 1 and cin % 64 == 0:
                for tile in ((128, 128), (64, 128)):
                    try:
                        runs["halo %d" % tile[0]] = ops.bench_conv(B * bm, cin, hw, hw, cout, k, st, pad, dil, tile=tile,
                                                                   halo=True, **kw)
                    except Exception as e:  # noqa: BLE001
                        runs["halo %d" % tile[0]] = "ERR %s" % str(e)[:60]
            for tile in WREG:
                for stg in (3, 4):
                    try:
                        runs["wreg %dx%d s%d" % (tile[0], tile[1], stg)] = ops.bench_conv(
                            B * bm, cin, hw, hw, cout, k, st, pad, dil, tile=tile, stages=stg, wreg=True, **kw)
                    except Exception as e:  # noqa: BLE001
                        runs["wreg %dx%d s%d" % (tile[0], tile[1], stg)] = "ERR %s" % str(e)[:60]
            ok = {k_: v for k_, v in runs.items() if isinstance(v, float)}
            old = {k_: v for k_, v in ok.items() if not k_.startswith("wreg")}
            new = {k_: v for k_, v in ok.items() if k_.startswith("wreg")}
            bo, bn = min(old, key=old.get), (min(new, key=new.get) if new else None)
            res[B][name] = {"gflop": round(gflop, 3), "best_old": bo, "old_us": round(old[bo], 2),
                            "best_wreg": bn, "wreg_us": round(new[bn], 2) if bn else None,
                            "runs": {k_: (round(v, 2) if isinstance(v, float) else v) for k_, v in runs.items()}}
            print("B=%d %-12s %7.2f GF  old %-14s %7.2f us %6.0f TF | wreg %-16s %7.2f us %6.0f TF  x%.2f  [%.0fs]" % (
                B, name, gflop, bo, old[bo], gflop / old[bo] * 1e3, bn, new[bn] if bn else 0,
                gflop / new[bn] * 1e3 if bn else 0, old[bo] / new[bn] if bn else 0, time.time() - t0), flush=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
