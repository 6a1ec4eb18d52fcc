#!/usr/bin/env python
"""per-launch times of the Refine part (eager, HIP events) with the chain kernel on / off"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from siammask_amd import _lib, synth
from siammask_amd.custom import build
dev = torch.device("cuda", 0)
sd = synth.torch_state_dict("sharp", "synthetic_damped")
for B in (1, 8, 64):
    for chain in (0, 1):
        _lib.tune(chain=chain)
        m = build("sharp", dtype="f16", max_batch=B, graph=False)
        m.load_state_dict(sd); m = m.eval().to(dev)
        m.template(torch.from_numpy(synth.image_batch(B, 127, stream0=0)).to(dev))
        x = torch.from_numpy(synth.image_batch(B, 255, stream0=1000)).to(dev)
        tw = torch.full((B, 2), 70.0, device=dev)
        for _ in range(3): m.track_step(x, tw, refine=True)
        m.profile(True)
        for _ in range(5): m.track_step(x, tw, refine=True)
        torch.cuda.synchronize()
        recs = m.profile_dump(); m.profile(False)
        names = ("deconv", "v2.0", "v1.0", "v0.0", "v2.2", "v1.2", "v0.2", "h2.0", "h2.2", "post0", "h1.0", "h1.2", "post1",
                 "h0.0", "h0.2", "post2", "refine_chain")
        sel = [r for r in recs if r["id"] in names]
        tot = sum(r["ms"] / r["calls"] for r in sel) * 1e3
        print("B=%d chain=%d refine total %.1f us: " % (B, chain, tot) +
              " ".join("%s=%.1f" % (r["id"], r["ms"] / r["calls"] * 1e3) for r in sel), flush=True)
_lib.tune(chain=1)
