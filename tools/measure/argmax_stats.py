"""fp16 argmax agreement (north_star: "bit-exact for the argmax box index").

The fp32 device path is the pinned one (<= 6e-6 of the reference's outputs, best anchor index bit-exact against the fp64 oracle
and against the unchanged tools/test.py at B = 1, 2, 8, 64).  The fp16 path -- the dtype of the headline number -- stores
activations in fp16, so its `pscore` (tools/test.py:235-238) carries an error and its argmax (tools/test.py:239) can
legitimately land on another candidate whenever the two best candidates are closer than that error.  This module measures
how often, over many independent streams, using nothing but the product path:

    for every stream:  best_id of the fp16 context  vs  best_id of the fp32 context   (both decoded ON DEVICE, smk_step)
    at a mismatch:     gap  = pscore32[best32] - pscore32[best16]   (how much worse fp16's pick is in fp32's own ranking)
                       err  = max |pscore16 - pscore32| over the 3125 candidates of that stream

`gap <= 2 * err` must hold at every mismatch (the fp16 pick is the argmax of the fp16 pscore, so it cannot be further from
the fp32 winner than the two pscore errors) -- a violation would mean a decode or plumbing defect, not rounding.
The pscore vectors come from a host restatement handed in by the caller (`host_pscore`; tests pass oracle.np_oracle.decode_best,
bench.py passes None and reports the rates only: the product path never imports the oracle).

Used by tests/test_gpu_argmax.py (gate + fp64-oracle spot checks) and by bench.py (`argmax_agreement` in the JSON line).
"""
import numpy as np
import torch


def _model(variant, dtype, B, mode=None):
    from siammask_amd import synth
    from siammask_amd.custom import build
    m = build(variant, dtype=dtype, max_batch=B, graph=True)
    m.load_state_dict(synth.torch_state_dict(variant, "synthetic_damped"))
    return m.eval().cuda()


def collect(B=64, seeds=8, kinds=("smooth", "noise"), variant="sharp", host_pscore=None, keep_tensors=False, models=None):
    """-> dict(streams, agree, rate, per_kind, mismatches=[...]).  One template + one frame per stream."""
    from siammask_amd import synth
    m16, m32 = models or (_model(variant, "f16", B), _model(variant, "f32", B))
    gen = {"smooth": synth.smooth_image_batch, "noise": synth.image_batch}
    out = {"streams": 0, "agree": 0, "per_kind": {}, "mismatches": [], "batch": B, "seeds": seeds, "variant": variant,
           "best16": {}, "best32": {}}          # per kind: [seeds][B] device best_id of each context (the fp64-oracle comparison of the test)
    for kind in kinds:
        pk = out["per_kind"].setdefault(kind, {"streams": 0, "agree": 0})
        for seed in range(seeds):
            s0 = 10000 * (seed + 1)
            z = torch.from_numpy(gen[kind](B, 127, stream0=s0)).cuda()
            x = torch.from_numpy(gen[kind](B, 255, stream0=s0 + 5000)).cuda()
            g = np.random.Generator(np.random.PCG64(7 + seed))
            twh_h = g.uniform(40.0, 110.0, size=(B, 2))
            twh = torch.from_numpy(twh_h).cuda()
            res = []
            for m in (m16, m32):
                m.template(z)
                o = m.track_step(x, twh, refine=False, mask_head=False)
                res.append({k: o[k].cpu().numpy() for k in ("box", "cls", "loc")})
            torch.cuda.synchronize()
            b16, b32 = res[0]["box"][:, 7].astype(np.int64), res[1]["box"][:, 7].astype(np.int64)
            out["best16"].setdefault(kind, []).append(b16)
            out["best32"].setdefault(kind, []).append(b32)
            same = b16 == b32
            out["streams"] += B; out["agree"] += int(same.sum())
            pk["streams"] += B; pk["agree"] += int(same.sum())
            for b in np.nonzero(~same)[0]:
                rec = {"kind": kind, "seed": seed, "stream": int(b), "best16": int(b16[b]), "best32": int(b32[b]),
                       "pscore16_dev": float(res[0]["box"][b, 6]), "pscore32_dev": float(res[1]["box"][b, 6]),
                       "target_wh": twh_h[b].tolist()}
                if host_pscore is not None:
                    p32 = host_pscore(res[1]["cls"][b], res[1]["loc"][b], twh_h[b])
                    p16 = host_pscore(res[0]["cls"][b], res[0]["loc"][b], twh_h[b])
                    rec["gap"] = float(p32[b32[b]] - p32[b16[b]])
                    rec["err"] = float(np.abs(p16 - p32).max())
                    # how far the DEVICE's pick is below the host restatement's maximum of the same tensors (0 unless a near-tie
                    # meets a 1-ulp difference of expf: tests/test_gpu_dropin.py's near-tie property)
                    rec["tie32"] = float(p32.max() - p32[b32[b]])
                    rec["tie16"] = float(p16.max() - p16[b16[b]])
                    top2 = np.partition(p32, -2)[-2:]
                    rec["top2_gap32"] = float(top2[1] - top2[0])
                if keep_tensors:
                    rec["_z"], rec["_x"] = z[b].cpu().numpy(), x[b].cpu().numpy()
                out["mismatches"].append(rec)
    out["rate"] = out["agree"] / max(1, out["streams"])
    for pk in out["per_kind"].values():
        pk["rate"] = pk["agree"] / max(1, pk["streams"])
    return out


def summary(st):
    """the part that goes into the bench line"""
    mm = st["mismatches"]
    s = {"streams": st["streams"], "agree": st["agree"], "rate": round(st["rate"], 5),
         "per_kind": {k: round(v["rate"], 5) for k, v in st["per_kind"].items()},
         "what": "device best_id (tools/test.py:239) of the fp16 context == that of the fp32 context (the pinned path), "
                 "B=%d x %d seeds x %s inputs, %s" % (st["batch"], st["seeds"], "/".join(st["per_kind"]), st["variant"])}
    if mm and "gap" in mm[0]:
        s["mismatch_gap_median"] = float(np.median([m["gap"] for m in mm]))
        s["mismatch_err_median"] = float(np.median([m["err"] for m in mm]))
        s["mismatch_gap_over_2err_max"] = float(max(m["gap"] / (2 * m["err"]) for m in mm))
    return s
