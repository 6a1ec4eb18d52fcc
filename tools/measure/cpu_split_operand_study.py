"""CPU study (no GPU): would a SPLIT-OPERAND fp16 path -- activations and weights as fp16 hi + lo pairs, three MFMA
products per multiply-accumulate (hi*hi + hi*lo + lo*hi; the lo*lo term is dropped), fp32 accumulation -- reproduce the fp64 reference's
argmax box index (/root/reference/tools/test.py:237 np.argmax(pscore)) on the 1024 streams of tests/golden/argmax_oracle_1024.npz?
On the 2.5 PFLOP/s fp16 pipe that is 833 TFLOP/s effective, 5x the fp32-MFMA peak the only argmax-exact context runs on today.

Model (QuantOracle with q16 replaced):
  value v -> hi = fp16(v), lo = fp16((v - hi) * 2^11) / 2^11  (the 2^11 keeps lo out of the fp16 subnormals); what is STORED and what the
             next layer reads is hi + lo (two fp16 planes, the bytes of one fp32);
  conv     -> conv(x_hi, w_hi + w_lo) + conv(x_lo, w_hi): exactly the three products the kernel would issue (x_lo * w_lo never formed);
             sums exact here (fp32 on the device: the fp32 context, with the same accumulators, is 1024 / 1024);
  the rest -> as QuantOracle (BN folded in float64, fp32 bias, ReLU / residual / maxpool / dw-xcorr on the stored values).
Variants: "x3" as above; "x3u" = the same with the lo planes stored UNSCALED (fp16 subnormals carry them: absolute precision 2^-25) -- the form
a K-concatenated fp16 GEMM [x_hi | x_hi | x_lo] x [w_hi | w_lo | w_hi] can consume with the existing kernels; "x2w" = weights split, activations single fp16 (two products); "x2a" = activations split, weights single fp16:
they tell WHICH rounding flips the picks if x3 is not exact.

Usage: python tools/measure/cpu_split_operand_study.py [variant,...] [streams_per_(kind,seed) | "mism"] [out.json]
  "mism" = only the streams on which the fp16 context disagrees with the oracle (profiles/r05_argmax_agreement.json), the cheap first pass."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import np_oracle as O                      # noqa: E402
from oracle.make_argmax_golden import inputs, KINDS    # noqa: E402
from siammask_amd import synth                          # noqa: E402

_q16 = O.q16


LO_SCALE = 2048.0       # 2^11 keeps lo out of the fp16 subnormals; 1.0 = lo stored unscaled (subnormals: absolute precision 2^-25)


def split(a):
    hi = _q16(a)
    lo = _q16((np.asarray(a, np.float64) - hi) * LO_SCALE) / LO_SCALE
    return hi, lo


def qsplit(a):
    hi, lo = split(a)
    return hi + lo


class SplitOracle(O.QuantOracle):
    """see the module docstring; wsplit / asplit choose which operand carries a lo plane"""

    def __init__(self, sd, wsplit=True, asplit=True):
        super(SplitOracle, self).__init__(sd, "sharp")
        self.wsplit, self.asplit = wsplit, asplit

    def _fold(self, conv, bn=None):
        w = self.sd[conv + ".weight"]
        co = w.shape[0]
        scale, shift = np.ones(co), np.zeros(co)
        if bn is not None:
            s = self.sd
            scale = s[bn + ".weight"] / np.sqrt(s[bn + ".running_var"] + O.BN_EPS)
            shift = s[bn + ".bias"] - s[bn + ".running_mean"] * scale
        if conv + ".bias" in self.sd:
            shift = shift + self.sd[conv + ".bias"]
        wf = w * scale.reshape(-1, 1, 1, 1)
        return (qsplit(wf) if self.wsplit else _q16(wf)), shift.astype(np.float32).astype(np.float64)

    def _fused(self, x, conv, bn=None, stride=1, pad=0, dil=1, act=False, res=None, res_after_relu=False, store=True):
        w, b = self._fold(conv, bn)
        xh = _q16(x)
        y = O.conv2d(xh, w, b, stride, pad, dil)                       # x_hi * (w_hi + w_lo)
        if self.asplit:
            xl = np.asarray(x, np.float64) - xh
            y = y + O.conv2d(xl, _q16(w), None, stride, pad, dil)      # x_lo * w_hi
        if res is not None and not res_after_relu:
            y = y + res
        if act:
            y = O.relu(y)
        if res is not None and res_after_relu:
            y = y + res
        return self._store(y) if store else y

    def _store(self, y):
        return qsplit(y) if self.asplit else _q16(y)


def run_variant(name, streams, sd, golden):
    global LO_SCALE
    LO_SCALE = 1.0 if name.endswith("u") else 2048.0              # "x3u": the lo planes unscaled (what a K-concatenated fp16 GEMM can consume as is)
    wsplit, asplit = {"x3": (True, True), "x3u": (True, True), "x2w": (True, False), "x2a": (False, True), "f16": (False, False)}[name]
    o = SplitOracle(sd, wsplit, asplit)
    saved = O.q16
    O.q16 = (lambda a: qsplit(a)) if asplit else saved             # every other rounding point of QuantOracle (cvt_in, corr, ...)
    res = {"variant": name, "streams": 0, "agree": 0, "mismatches": []}
    t0 = time.time()
    try:
        by_chunk = {}
        for (ki, seed, b) in streams:
            by_chunk.setdefault((ki, seed, b // 8), []).append(b)
        for (ki, seed, c), bs in sorted(by_chunk.items()):
            z, x, twh = inputs(KINDS[ki], seed, c * 8, 8)
            sel = [b - c * 8 for b in bs]
            o.template(z[sel].astype(np.float64))
            cls, loc = o.track(x[sel].astype(np.float64))[:2]
            for j, b in enumerate(bs):
                ps = O.decode_best(cls[j], loc[j], target_sz=twh[sel[j]], scale_x=1.0)[3]
                best = int(np.argmax(ps))
                ref = int(golden["top_idx"][ki, seed, b, 0])
                res["streams"] += 1
                if best == ref:
                    res["agree"] += 1
                else:
                    gap = float(golden["top_pscore"][ki, seed, b, 0] - golden["top_pscore"][ki, seed, b, 1])
                    res["mismatches"].append({"kind": KINDS[ki], "seed": seed, "stream": b, "pick": best, "oracle": ref,
                                              "oracle_top2_gap": gap,
                                              "pick_rank_in_oracle_top5": int(np.where(golden["top_idx"][ki, seed, b] == best)[0][0])
                                              if best in golden["top_idx"][ki, seed, b] else -1})
            print("%s: %d / %d agree (%.0f s)" % (name, res["agree"], res["streams"], time.time() - t0), flush=True)
    finally:
        O.q16 = saved
    res["rate"] = res["agree"] / max(1, res["streams"])
    res["seconds"] = round(time.time() - t0, 1)
    return res


def main():
    variants = (sys.argv[1] if len(sys.argv) > 1 else "x3").split(",")
    what = sys.argv[2] if len(sys.argv) > 2 else "mism"
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(REPO, "profiles", "r06_cpu_split_operand_study.json")
    golden = np.load(os.path.join(REPO, "tests", "golden", "argmax_oracle_1024.npz"))
    sd = synth.state_dict("sharp", "synthetic_damped")
    if what == "mism":
        prof = json.load(open(os.path.join(REPO, "profiles", "r05_argmax_agreement.json")))
        streams = sorted({(KINDS.index(m["kind"]), int(m["seed"]), int(m["stream"])) for m in prof["mismatches"] if m.get("dtype", "f16") == "f16"})
    else:
        n = int(what)
        streams = [(ki, seed, b) for ki in range(2) for seed in range(8) for b in range(n)]
    results = {"what": "argmax box index of split-operand fp16 variants (CPU model, exact sums) vs the fp64 oracle's "
                       "(tests/golden/argmax_oracle_1024.npz)", "selection": what, "n_streams": len(streams), "variants": []}
    for v in variants:
        results["variants"].append(run_variant(v, streams, sd, golden))
        with open(out, "w") as f:
            json.dump(results, f, indent=1)
    print(json.dumps({v["variant"]: [v["agree"], v["streams"]] for v in results["variants"]}))


if __name__ == "__main__":
    main()
