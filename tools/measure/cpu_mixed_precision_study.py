"""CPU study (no GPU): would fp32 STORAGE of parts of the fp16 path buy argmax agreement with the fp64 reference?
("one experiment: fp32 storage for the residual trunk (Y of every Bottleneck) or for search -> heads only")

Variants of oracle.np_oracle.QuantOracle (fp16 folded weights, fp16 stored activations, exact sums):
  f16        every stored activation fp16 (= the device's fp16 mode)
  trunk32    the Bottleneck outputs (the residual trunk: conv3 + shortcut sums, p1 / p2 / p3) are kept unrounded; every
             convolution still reads fp16 operands (the trunk is rounded when it is used as a conv input, as an MFMA must)
  heads32    `search` and everything behind it (conv_search / conv_kernel, xcorr, head.0) unrounded, backbone as f16
  both       trunk32 + heads32
For N streams per input kind: best_id of each variant vs the fp64 Oracle's, pscore error, top-2 gap.
Usage: python tools/measure/cpu_mixed_precision_study.py [streams_per_kind] [out.json]"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import np_oracle as O          # noqa: E402
from siammask_amd import synth             # noqa: E402


class MixedOracle(O.QuantOracle):
    def __init__(self, sd, trunk32=False, heads32=False):
        super(MixedOracle, self).__init__(sd, "sharp")
        self.trunk32, self.heads32 = trunk32, heads32
        self._in_heads = False

    def _fused(self, x, conv, bn=None, stride=1, pad=0, dil=1, act=False, res=None, res_after_relu=False, store=True):
        if self._in_heads and self.heads32:
            store = False
        return super(MixedOracle, self)._fused(O.q16(x) if self.trunk32 else x, conv, bn, stride, pad, dil, act, res,
                                               res_after_relu, store)

    def _bottleneck(self, x, p, stride, dil, ds):
        if not self.trunk32:
            return super(MixedOracle, self)._bottleneck(x, p, stride, dil, ds)
        pad2 = dil if dil > 1 else 2 - stride
        out = self._fused(x, p + "conv1", p + "bn1", act=True)
        out = self._fused(out, p + "conv2", p + "bn2", stride, pad2, dil, act=True)
        if ds is not None:
            k, s, pd = ds
            residual = self._fused(x, p + "downsample.0", p + "downsample.1", s, pd, 1, store=False)
        else:
            residual = x
        return self._fused(out, p + "conv3", p + "bn3", act=True, res=residual, store=False)

    def resdown(self, x):
        feats = self.resnet(x)
        d = "features.downsample.downsample."
        self._in_heads = True
        y = self._fused(feats[3], d + "0", d + "1")
        self._in_heads = False
        if y.shape[3] < 20:
            y = y[:, :, 4:-4, 4:-4]
        return feats, y

    def forward_corr(self, p, kernel, search):
        self._in_heads = True
        k = self._fused(kernel, p + "conv_kernel.0", p + "conv_kernel.1", act=True)
        s = self._fused(search, p + "conv_search.0", p + "conv_search.1", act=True)
        corr = O.conv2d_dw_group(s, k)
        if not self.heads32:
            corr = O.q16(corr)
        self._in_heads = False
        return corr

    def head(self, p, feature):
        self._in_heads = True
        h = self._fused(feature, p + "head.0", p + "head.1", act=True)
        self._in_heads = False
        return self._fused(h, p + "head.3", store=False)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles", "r04_cpu_mixed_precision_study.json")
    sd = synth.state_dict("sharp", "synthetic_damped")
    ref = O.Oracle(sd, "sharp")
    var = {"f16": MixedOracle(sd), "trunk32": MixedOracle(sd, trunk32=True), "heads32": MixedOracle(sd, heads32=True),
           "both": MixedOracle(sd, True, True)}
    rows = []
    t0 = time.time()
    for kind, gen in (("smooth", synth.smooth_image_batch), ("noise", synth.image_batch)):
        for b0 in range(0, n, 4):
            B = min(4, n - b0)
            z = gen(B, 127, stream0=20000 + b0).astype(np.float64)
            x = gen(B, 255, stream0=25000 + b0).astype(np.float64)
            g = np.random.Generator(np.random.PCG64(17 + b0))
            twh = g.uniform(40.0, 110.0, size=(B, 2))
            ref.template(z)
            rc, rl = ref.track(x)
            res = {}
            for name, o in var.items():
                o.template(z)
                res[name] = o.track(x)
            for b in range(B):
                bid, _, _, ps = O.decode_best(rc[b].astype(np.float32), rl[b].astype(np.float32), target_sz=twh[b], scale_x=1.0)
                top2 = np.partition(ps, -2)[-2:]
                row = {"kind": kind, "stream": b0 + b, "best_ref": bid, "top2_gap": float(top2[1] - top2[0])}
                for name, (c, l) in res.items():
                    vb, _, _, vps = O.decode_best(c[b].astype(np.float32), l[b].astype(np.float32), target_sz=twh[b], scale_x=1.0)
                    row[name] = {"best": vb, "agree": vb == bid, "pscore_err": float(np.abs(vps - ps).max()),
                                 "cls_err": float(np.abs(c[b] - rc[b]).max() / np.abs(rc[b]).max())}
                rows.append(row)
            print("%s %d/%d  %.0f s" % (kind, b0 + B, n, time.time() - t0), flush=True)
    summ = {name: {"agree": int(sum(r[name]["agree"] for r in rows)), "streams": len(rows),
                   "median_pscore_err": float(np.median([r[name]["pscore_err"] for r in rows])),
                   "median_cls_err": float(np.median([r[name]["cls_err"] for r in rows]))} for name in var}
    summ["median_top2_gap_ref"] = float(np.median([r["top2_gap"] for r in rows]))
    json.dump({"summary": summ, "rows": rows}, open(out, "w"), indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
