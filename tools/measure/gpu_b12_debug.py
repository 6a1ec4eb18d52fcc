#!/usr/bin/env python
"""Round 3: B = 12 with the persistent sequence differs from the per-launch path by 1e-1 on cls (test_which_batches...[12]) --
which image, which tensor, which feature of the sequence?  Per-image relative differences of p2 / p3 / search / cls against the
per-launch path for the combinations of seq_fuse and seq_halo."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from siammask_amd import _lib, synth
from siammask_amd.custom import build

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=11)).cuda()
x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=11)).cuda()
twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()


def run(**knobs):
    _lib.tune(**knobs)
    m = build("sharp", dtype="f16", graph=True, max_batch=B)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    m = m.eval().cuda()
    m.template(z)
    zf = m.debug_tensor("zf").cpu().numpy().astype(np.float64)
    out = m.track_step(x, twh, refine=True)
    d = {k: out[k].cpu().numpy().astype(np.float64) for k in ("cls", "loc")}
    d["zf"] = zf
    for n in ("p2", "p3", "search"):
        d[n] = m.debug_tensor(n).cpu().numpy().astype(np.float64)
    torch.cuda.synchronize()
    st = m.seq_status()
    del m
    return d, st


base, _ = run(seq=0)
for knobs in (dict(seq=1, seq_extra_batch=B, seq_fuse=3), dict(seq=1, seq_extra_batch=B, seq_fuse=3), dict(seq=1, seq_extra_batch=B, seq_fuse=3)):
    d, st = run(**knobs)
    print("== %s  status %s  fused pairs (last launch) %d" % (knobs, st, _lib.tune_get("seq_fused_last")))
    for n in ("p2", "p3"):
        per = [float(np.abs(d[n][b] - base[n][b]).max() / (np.abs(base[n][b]).max() + 1e-30)) for b in range(B)]
        print("   %-7s per image: %s" % (n, " ".join("%.1e" % v for v in per)))
        for b in range(B):
            if per[b] > 1e-2 and n == "p2":
                e = np.abs(d[n][b] - base[n][b]) / (np.abs(base[n][b]).max() + 1e-30)      # [C, H, W]
                C = e.shape[0]
                px = e.reshape(C, -1)
                bad_px = np.nonzero(px.max(axis=0) > 1e-2)[0]
                bad_ch = np.nonzero(px.max(axis=1) > 1e-2)[0]
                print("      image %d %s: %d bad pixels of %d: 32-pixel blocks %s; image rows %s" % (
                    b, n, len(bad_px), px.shape[1], sorted(set((bad_px // 32).tolist())), sorted(set((bad_px // 31).tolist()))))
                print("      first bad pixels %s" % bad_px[:24].tolist())
                print("      bad channels: %d of %d; 64-channel blocks %s; first %s" % (
                    len(bad_ch), C, sorted(set((bad_ch // 64).tolist())), bad_ch[:12].tolist()))
                worst = np.unravel_index(np.argmax(e), e.shape)
                print("      worst at (c, y, x) = %s: got %.4f want %.4f" % (worst, d[n][b][worst], base[n][b][worst]))
_lib.tune(seq=1, seq_extra_batch=0, seq_fuse=1)
