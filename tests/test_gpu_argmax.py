"""fp16 argmax parity as a number and a gate (north_star: "bit-exact for the argmax box index";
/root/reference/tools/test.py:237-239,253).

fp32 is the pinned path: its best anchor index equals the fp64 oracle's (and the unchanged tool's) everywhere it is
checked.  fp16 -- the dtype of the headline bench number -- may pick another candidate when the two best candidates of a
stream are closer than the fp16 error of `pscore`.  Over 1024 streams (B = 64 x 8 seeds x smooth / white-noise inputs):

  * agreement rate of the device-decoded best_id, fp16 context vs fp32 context -> gpurun_out/argmax_agreement.json (bench.py
    prints the same statistic as `argmax_agreement`);
  * gate: at EVERY mismatch the fp16 pick is within 2 x (measured fp16 pscore error of that stream) of the fp32 winner in
    fp32's own ranking, and the device's picks are the host restatement's picks on the device's own cls / loc;
  * fp64-oracle spot checks: on mismatching streams (and two agreeing ones) the fp32 device index equals the index the
    numpy oracle computes from scratch in float64 -- the fp32 column of the comparison really is the reference's answer.
  * round 5: BOTH device dtypes against the fp64 oracle on ALL 1024 streams.  The oracle's five best
    candidates per stream were computed on the CPU (oracle/make_argmax_golden.py, 25 min of an 8-core host -- not GPU-box
    time) and are committed as tests/golden/argmax_oracle_1024.npz.  fp32: the device index IS the oracle's on every stream,
    up to genuine float32 near-ties (the pick's float64 pscore within 2e-5 of the oracle's maximum; counted and reported).
    fp16: the agreement rate is reported and held to a hard floor (the measured rate minus one point).
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle.np_oracle import Oracle, decode_best
from siammask_amd import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools", "measure"))
OUT = os.path.join(REPO, "gpurun_out")


FP16_VS_ORACLE_FLOOR = 0.971      # measured 0.98145 (1005 / 1024, profiles/r05_argmax_agreement.json) minus one point


def _host_pscore(cls, loc, twh):
    return decode_best(cls, loc, target_sz=twh, scale_x=1.0)[3]


def test_fp16_argmax_agreement_rate_and_gate():
    import argmax_stats
    st = argmax_stats.collect(B=64, seeds=8, kinds=("smooth", "noise"), host_pscore=_host_pscore, keep_tensors=True)
    s = argmax_stats.summary(st)
    mm = st["mismatches"]
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "argmax_agreement.json"), "w") as f:
        json.dump({"summary": s, "mismatches": [{k: v for k, v in m.items() if not k.startswith("_")} for m in mm]}, f, indent=1)
    print("fp16 vs fp32 device argmax: %d / %d streams agree (%.3f %%); per kind %s; %d mismatches, median gap %.3g, median "
          "pscore error %.3g" % (st["agree"], st["streams"], 100 * st["rate"], s["per_kind"], len(mm),
                                 s.get("mismatch_gap_median", 0.0), s.get("mismatch_err_median", 0.0)))
    assert st["streams"] >= 512
    for m in mm:
        # the device decode and the host restatement agree on both dtypes' own tensors ...
        assert m["tie32"] <= 1e-6 and m["tie16"] <= 1e-6, m
        # ... and the fp16 pick is explained by the fp16 pscore error of that stream
        assert -1e-6 <= m["gap"] <= 2.0 * m["err"] + 1e-6, "fp16 picked a candidate %.3g below the fp32 winner with a pscore error of only %.3g: %s" % (
            m["gap"], m["err"], {k: v for k, v in m.items() if not k.startswith("_")})
    # fp64 oracle from scratch on up to four mismatching streams: the fp32 column is the reference's answer
    o = Oracle(synth.state_dict("sharp", "synthetic_damped"), "sharp")
    for m in mm[:4]:
        o.template(m["_z"][None].astype(np.float64))
        ocls, oloc = o.track(m["_x"][None].astype(np.float64))[:2]
        bid = decode_best(ocls[0], oloc[0], target_sz=np.asarray(m["target_wh"]), scale_x=1.0)[0]
        assert bid == m["best32"], "fp32 device index %d != fp64 oracle %d (fp16 picked %d)" % (m["best32"], bid, m["best16"])
    # ---- both dtypes against the fp64 oracle, all 1024 streams (fixture: oracle/make_argmax_golden.py) ----------------------
    gold = np.load(os.path.join(REPO, "tests", "golden", "argmax_oracle_1024.npz"))
    kinds = [str(k) for k in gold["kinds"]]
    top_idx, top_ps = gold["top_idx"].astype(np.int64), gold["top_pscore"]
    vs = {}
    for name in ("best32", "best16"):
        exact = near = total = 0
        worst = 0.0
        per_kind = {}
        for ki, kind in enumerate(kinds):
            got = np.stack(st[name][kind])                                   # [seeds][B]
            want = top_idx[ki, :, :, 0]
            same = got == want
            # a pick that is not the oracle's: how far below the oracle's maximum is it in the ORACLE's float64 ranking?
            hit = got[..., None] == top_idx[ki]                              # [seeds][B][5]
            ps_pick = np.where(hit.any(-1), (top_ps[ki] * hit).sum(-1), -np.inf)
            deficit = top_ps[ki, :, :, 0] - ps_pick
            exact += int(same.sum()); total += same.size
            near += int((~same & (deficit <= 2e-5)).sum())
            worst = max(worst, float(np.where(same, 0.0, np.where(np.isfinite(deficit), deficit, 1.0)).max()))
            per_kind[kind] = round(float(same.mean()), 5)
        vs[name] = {"streams": total, "exact": exact, "rate": round(exact / total, 5), "float_near_ties": near,
                    "worst_deficit_in_oracle_ranking": worst, "per_kind": per_kind}
    with open(os.path.join(OUT, "argmax_agreement.json"), "w") as f:
        json.dump({"summary": s, "vs_fp64_oracle": {"fp32": vs["best32"], "fp16": vs["best16"]},
                   "mismatches": [{k: v for k, v in m.items() if not k.startswith("_")} for m in mm]}, f, indent=1)
    print("vs the fp64 oracle, 1024 streams: fp32 %s ; fp16 %s" % (vs["best32"], vs["best16"]))
    assert vs["best32"]["streams"] == 1024
    # fp32 (north_star: "bit-exact for the argmax box index"): the oracle's index everywhere, float32 near-ties excepted
    assert vs["best32"]["exact"] + vs["best32"]["float_near_ties"] == 1024, vs["best32"]
    assert vs["best32"]["exact"] >= 1022, vs["best32"]          # (measured: 1024 exact, no near-tie)
    # fp16, the headline dtype: hard floors at the measured rates minus one point (round 5, profiles/r05_argmax_agreement.json:
    # 0.981 vs the fp32 context and, the fp32 context being the oracle's index on all 1024 streams, 0.981 vs the oracle) -- the rates are a property of the synthetic checkpoint (no trained
    # attractor; the net amplifies perturbations ~27x), a drop below them is a regression of the fp16 path
    assert st["rate"] >= 0.971, s
    assert vs["best16"]["rate"] >= FP16_VS_ORACLE_FLOOR, vs["best16"]
