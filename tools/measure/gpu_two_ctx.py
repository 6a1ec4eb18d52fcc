#!/usr/bin/env python
"""Software pipelining across frames: two contexts (own arena, graphs, HIP stream) take alternate frame batches, so that the
chip-idle Refine tail of frame k (decode -> window convs -> chain, ~100 us with 8..40 workgroups) can overlap the HBM-bound
head (cvt_in, stem, maxpool, layer1) of frame k+1.  Compares frames/s with one context on one stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from siammask_amd import synth
from siammask_amd.custom import build


def make(B):
    m = build("sharp", dtype="f16", max_batch=B, graph=True)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    return m.eval().cuda()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = 200
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=3)).cuda()
    xs = [torch.from_numpy(synth.image_batch(B, 255, stream0=1000 * (i + 1))).cuda() for i in range(4)]
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
    for nctx in (1, 2, 3, 1, 2):
        ms = [make(B) for _ in range(nctx)]
        ss = [torch.cuda.Stream() for _ in range(nctx)]
        for m, s in zip(ms, ss):
            with torch.cuda.stream(s):
                m.template(z)
                for i in range(10):
                    m.track_step(xs[i % 4], twh, refine=True, stage=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % nctx
            with torch.cuda.stream(ss[k]):
                ms[k].track_step(xs[i % 4], twh, refine=True, stage=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = [m.seq_status() for m in ms]
        print("B=%d contexts=%d  %.4f ms/step  %.0f frames/s   seq %s" % (B, nctx, dt / steps * 1e3, B * steps / dt, st), flush=True)
        del ms


if __name__ == "__main__":
    main()
