#!/usr/bin/env python
"""Round 3, open item (profiles/r03h_b12_race.txt): which batches / which teams show it?  The sequence and the pair fusion forced
(seq_min_batch 1 .. seq_max_batch 64, seq_fuse 3) for B = 9 .. 16, p2 per image against the per-launch path; optionally with
eager launches (argv[2] = eager) or with the template on the per-launch path (argv[2] = tpl_off)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import _lib, synth
from siammask_amd.custom import build

mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
FORCE = dict(seq_min_batch=1, seq_max_batch=64, seq_mult_max=64)


def run(B, seq, **knobs):
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=11)).cuda()
    x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=11)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
    _lib.tune(seq=seq, **FORCE)
    _lib.tune(**knobs)
    m = build("sharp", dtype="f16", graph=(mode != "eager"), max_batch=B)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    m = m.eval().cuda()
    if mode == "tpl_off":
        _lib.tune(seq=0)
    m.template(z)
    _lib.tune(seq=seq)
    m.track_step(x, twh, refine=True)
    p2 = m.debug_tensor("p2").cpu().numpy().astype(np.float64)
    torch.cuda.synchronize()
    st = m.seq_status()
    del m
    return p2, st


for B in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "9,10,11,12,16").split(",")]:
    base, _ = run(B, 0)
    for rep in range(2):
        p2, st = run(B, 1, seq_fuse=3)
        per = [float(np.abs(p2[b] - base[b]).max() / (np.abs(base[b]).max() + 1e-30)) for b in range(B)]
        print("B=%-2d %s rep %d status %s  p2 per image: %s   -> bad images %s" % (
            B, mode, rep, st, " ".join("%.0e" % v for v in per), [b for b in range(B) if per[b] > 1e-2]), flush=True)
_lib.tune(seq=1, seq_fuse=1, seq_min_batch=5, seq_max_batch=8, seq_mult_max=24)
