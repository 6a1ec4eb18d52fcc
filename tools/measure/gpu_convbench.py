#!/usr/bin/env python
"""Micro-benchmark of the MFMA conv kernel on every distinct convolution geometry of the path
(per-launch microseconds for each tile x K-tile x ring-depth instantiation), through the C ABI
(smk_bench_conv).  Output: JSON {batch: {layer: {"gflop":…, "runs": {"128x128/128/s3": us, …}}}}.
Used to derive the tile heuristics in conv_igemm.hip::choose_tile (results under profiles/)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: F401

from siammask_amd import ops

# name: (Cin, H, Cout, k, stride, pad, dil, res, nchw, win, pos_mul, pos_add, batch_mul)
LAYERS = {
    "stem":        (3, 255, 64, 7, 2, 0, 1, 0, 0, None, 0, 0, 1),
    "l1.0.ds":     (64, 63, 256, 1, 1, 0, 1, 0, 0, None, 0, 0, 1),
    "l1.0.c1":     (64, 63, 64, 1, 1, 0, 1, 0, 0, None, 0, 0, 1),
    "l1.c1":       (256, 63, 64, 1, 1, 0, 1, 0, 0, None, 0, 0, 1),
    "l1.c2":       (64, 63, 64, 3, 1, 1, 1, 0, 0, None, 0, 0, 1),
    "l1.c3":       (64, 63, 256, 1, 1, 0, 1, 1, 0, None, 0, 0, 1),
    "l2.0.ds":     (256, 63, 512, 3, 2, 0, 1, 0, 0, None, 0, 0, 1),
    "l2.0.c1":     (256, 63, 128, 1, 1, 0, 1, 0, 0, None, 0, 0, 1),
    "l2.0.c2":     (128, 63, 128, 3, 2, 0, 1, 0, 0, None, 0, 0, 1),
    "l2.c1":       (512, 31, 128, 1, 1, 0, 1, 0, 0, None, 0, 0, 1),
    "l2.c2":       (128, 31, 128, 3, 1, 1, 1, 0, 0, None, 0, 0, 1),
    "l2.c3":       (128, 31, 512, 1, 1, 0, 1, 1, 0, None, 0, 0, 1),
    "l3.0.ds":     (512, 31, 1024, 3, 1, 1, 1, 0, 0, None, 0, 0, 1),
    "l3.0.c1":     (512, 31, 256, 1, 1, 0, 1, 0, 0, None, 0, 0, 1),
    "l3.c1":       (1024, 31, 256, 1, 1, 0, 1, 0, 0, None, 0, 0, 1),
    "l3.c2":       (256, 31, 256, 3, 1, 2, 2, 0, 0, None, 0, 0, 1),
    "l3.c3":       (256, 31, 1024, 1, 1, 0, 1, 1, 0, None, 0, 0, 1),
    "conv_search": (256, 31, 768, 3, 1, 0, 1, 0, 0, None, 0, 0, 1),
    "head0":       (256, 25, 256, 1, 1, 0, 1, 0, 0, None, 0, 0, 3),
    "mask3":       (256, 25, 3969, 1, 1, 0, 1, 0, 1, None, 0, 0, 1),
    "cls3":        (256, 25, 10, 1, 1, 0, 1, 0, 1, None, 0, 0, 1),
    "v2.0":        (512, 31, 128, 3, 1, 1, 1, 0, 0, (15, 15), 1, -4, 1),
    "v1.0":        (256, 63, 64, 3, 1, 1, 1, 0, 0, (31, 31), 2, -8, 1),
    "v0.0":        (64, 125, 16, 3, 1, 1, 1, 0, 0, (61, 61), 4, -16, 1),
    "h2":          (32, 15, 32, 3, 1, 1, 1, 0, 0, None, 0, 0, 1),
    "v2.2":        (128, 15, 32, 3, 1, 1, 1, 1, 0, None, 0, 0, 1),
    "h1":          (16, 31, 16, 3, 1, 1, 1, 0, 0, None, 0, 0, 1),
    "v1.2":        (64, 31, 16, 3, 1, 1, 1, 1, 0, None, 0, 0, 1),
    "h0":          (4, 61, 4, 3, 1, 1, 1, 0, 0, None, 0, 0, 1),
    "v0.2":        (16, 61, 4, 3, 1, 1, 1, 1, 0, None, 0, 0, 1),
}
CONFIGS = [((128, 128), 128), ((128, 128), 256), ((128, 64), 128), ((128, 64), 256),
           ((64, 128), 128), ((64, 128), 256), ((64, 64), 256), ((256, 128), 128)]


def main():
    batches = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8]
    out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/convbench.json"
    dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
    only = set(sys.argv[4].split(",")) if len(sys.argv) > 4 else None
    res = {}
    t0 = time.time()
    for B in batches:
        res[B] = {}
        for name, (cin, hw, cout, k, st, pad, dil, r, nchw, win, pm, pa, bm) in LAYERS.items():
            if only and name not in only:
                continue
            Hl = win[0] if win else hw
            Ho = (Hl + 2 * pad - dil * (k - 1) - 1) // st + 1
            gflop = 2.0 * B * bm * Ho * Ho * cout * cin * k * k / 1e9
            runs = {}
            for tile, kt in CONFIGS:
                for stages in (2, 3, 4):
                    if (tile[0] + tile[1]) * kt * stages > 160 * 1024:
                        continue
                    key = "%dx%d/%d/s%d" % (tile[0], tile[1], kt, stages)
                    try:
                        runs[key] = round(ops.bench_conv(B * bm, cin, hw, hw, cout, k, st, pad, dil, dtype=dtype,
                                                         tile=tile, kt=kt, stages=stages, res=bool(r), nchw=bool(nchw),
                                                         win=win, pos_mul=pm, pos_add=pa, iters=30), 2)
                    except Exception as e:  # noqa: BLE001
                        runs[key] = "ERR %s" % str(e)[:80]
            ok = {k_: v for k_, v in runs.items() if isinstance(v, float)}
            best = min(ok, key=ok.get) if ok else None
            res[B][name] = {"gflop": round(gflop, 3), "best": best, "best_us": ok.get(best),
                            "best_tflops": round(gflop / ok[best] * 1e3, 1) if best else None, "runs": runs}
            print("B=%d %-12s %8.2f GF  best %-16s %8.2f us  %7.1f TF/s   [%.0fs]" % (
                B, name, gflop, best, ok.get(best, 0), gflop / ok[best] * 1e3 if best else 0, time.time() - t0), flush=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
