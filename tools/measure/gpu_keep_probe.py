#!/usr/bin/env python
"""What do the per-step result copies of bench.py (box [B,8] and fp16 mask [B,16129] into the retention
buffers, two torch kernels between graph replays) cost?  Same process, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
import bench
from siammask_amd import spec
dev = torch.device("cuda", 0)
for name in ("sharp_b8_f16", "sharp_b1_f16", "sharp_b64_f16"):
    w = bench.Workload(name, dev, 0)
    steps = 200 if w.B < 64 else 60
    res_masks = torch.empty((w.B, steps, spec.REFINE_OUT ** 2), dtype=torch.float16, device=dev)
    res_box = torch.empty((w.B, steps, 8), dtype=torch.float32, device=dev)
    res_masks_t = torch.empty((steps, w.B, spec.REFINE_OUT ** 2), dtype=torch.float16, device=dev)
    res_box_t = torch.empty((steps, w.B, 8), dtype=torch.float32, device=dev)
    bench.prewarm(w, 1.0)
    row = []
    for rep in range(2):
        for mode in ("none", "strided", "contig"):
            for i in range(20): w.step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                box, loc, mask, ref = w.step(i)
                if mode == "strided":
                    res_box[:, i].copy_(box); res_masks[:, i].copy_(ref)
                elif mode == "contig":
                    res_box_t[i].copy_(box); res_masks_t[i].copy_(ref)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / steps * 1e3)
    print("%-14s no copies %.4f %.4f | [B,T,..] slices %.4f %.4f | [T,B,..] rows %.4f %.4f ms/step" %
          (name, row[0], row[3], row[1], row[4], row[2], row[5]), flush=True)
