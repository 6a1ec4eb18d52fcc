#!/usr/bin/env python
"""conv_wreg_kernel with two settings of a producer knob -- by default the activation rows by LDS-DMA (a_stage=0, "a0" /
"dma" below) against the register-staged rows (a_stage=1, "a1" / "regs"); AB_KNOB=npw AB_VALS=2,4 compares two against
four producer waves -- per layer geometry and workgroup shape, beside the best LDS-staged instantiations (conv_igemm tiles,
conv3x3_halo) of the same layer.
    python tools/measure/gpu_producer_bench.py 8[,64] gpurun_out/astage_bench.json
One line per layer: us per launch of every candidate (30-launch averages, one process)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: F401

from siammask_amd import _lib, ops
from gpu_convbench import LAYERS

ONLY = ("l1.c1", "l1.c2", "l1.c3", "l2.0.ds", "l2.0.c1", "l2.0.c2", "l2.c1", "l2.c2", "l2.c3", "l3.0.ds", "l3.0.c1", "l3.c1",
        "l3.c2", "l3.c3", "conv_search", "head0", "v2.0", "v1.0")
BASE = [((128, 128), 128, 2), ((64, 128), 128, 3), ((128, 64), 128, 3)]
WREG = [(64, 256), (64, 128), (64, 64), (128, 256), (128, 128), (128, 64)]


KNOB = os.environ.get("AB_KNOB", "a_stage")               # the producer knob under test and its two values
VALS = [int(v) for v in os.environ.get("AB_VALS", "0,1").split(",")]


def main():
    batches = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8]
    out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/astage_bench.json"
    a0 = _lib.tune_get(KNOB)
    res, t0 = {}, time.time()
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
                for a in (0, 1):
                    _lib.tune(**{KNOB: VALS[a]})
                    try:
                        runs["wreg %dx%d a%d" % (tile[0], tile[1], a)] = ops.bench_conv(
                            B * bm, cin, hw, hw, cout, k, st, pad, dil, tile=tile, stages=3, wreg=True, **kw)
                    except Exception as e:  # noqa: BLE001
                        runs["wreg %dx%d a%d" % (tile[0], tile[1], a)] = "ERR %s" % str(e)[:60]
            _lib.tune(**{KNOB: a0})
            ok = {k_: v for k_, v in runs.items() if isinstance(v, float)}
            old = {k_: v for k_, v in ok.items() if not k_.startswith("wreg")}
            w0 = {k_: v for k_, v in ok.items() if k_.endswith("a0")}
            w1 = {k_: v for k_, v in ok.items() if k_.endswith("a1")}
            bo, b0, b1 = min(old, key=old.get), min(w0, key=w0.get), min(w1, key=w1.get)
            t0n, t1n = "%s=%d" % (KNOB, VALS[0]), "%s=%d" % (KNOB, VALS[1])
            res[B][name] = {"gflop": round(gflop, 3), "best_lds_staged": bo, "lds_staged_us": round(old[bo], 2),
                            "knob": KNOB, "a0": VALS[0], "a1": VALS[1],
                            "best_wreg_a0": b0, "wreg_a0_us": round(w0[b0], 2),
                            "best_wreg_a1": b1, "wreg_a1_us": round(w1[b1], 2),
                            "runs": {k_: (round(v, 2) if isinstance(v, float) else v) for k_, v in runs.items()}}
            print("B=%d %-12s %7.2f GF | lds-staged %-14s %7.2f | wreg %s %-8s %7.2f | wreg %s %-8s %7.2f us  x%.2f  [%.0fs]" % (
                B, name, gflop, bo, old[bo], t0n, b0[5:-3], w0[b0], t1n, b1[5:-3], w1[b1], w0[b0] / w1[b1],
                time.time() - t0), flush=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
