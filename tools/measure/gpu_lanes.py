#!/usr/bin/env python
"""Does splitting a batch of streams into L independent lanes (own context, own HIP stream, own graph)
overlap the latency-bound launches?  total batch fixed; per-step join vs free-running."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from siammask_amd import synth
from siammask_amd.custom import build
dev = torch.device("cuda", 0)
sd = synth.torch_state_dict("sharp", "synthetic_damped")

def run(total_b, lanes, steps, join):
    b = total_b // lanes
    ms, xs, tws = [], [], []
    for l in range(lanes):
        m = build("sharp", dtype="f16", max_batch=b, graph=True)
        m.load_state_dict(sd)
        m = m.eval().to(dev)
        m.template(torch.from_numpy(synth.image_batch(b, 127, stream0=l * b)).to(dev))
        ms.append(m)
        xs.append(torch.from_numpy(synth.image_batch(b, 255, stream0=1000 + l * b)).to(dev))
        tws.append(torch.full((b, 2), 70.0, device=dev))
    sts = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
    main = torch.cuda.current_stream(dev)
    def step():
        if join:
            ev = torch.cuda.Event(); ev.record(main)
        for l in range(lanes):
            if join: sts[l].wait_event(ev)
            with torch.cuda.stream(sts[l]):
                ms[l].track_step(xs[l], tws[l], refine=True, stage=False)
            if join:
                e2 = torch.cuda.Event(); e2.record(sts[l]); main.wait_event(e2)
    for _ in range(15): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for total_b, lane_set, steps in ((8, (1, 2, 4, 8), 100), (64, (1, 2, 4, 8), 30), (16, (1, 2, 4), 60), (2, (1, 2), 100)):
    for join in (True, False):
        row = []
        for rep in range(2):
            for L in lane_set:
                row.append(run(total_b, L, steps, join))
        n = len(lane_set)
        print("B=%-3d %-8s " % (total_b, "join" if join else "free") +
              " | ".join("L=%d: %.3f %.3f ms" % (lane_set[i], row[i], row[n + i]) for i in range(n)), flush=True)
