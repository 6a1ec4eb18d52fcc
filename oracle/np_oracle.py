"""CPU oracle: numpy restatement of the SiamMask per-frame inference path.

    *** TEST INFRASTRUCTURE ONLY ***
    Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
    module, and only as the checker.  The product path (siammask_amd/) never imports it
    and fails loudly when the HIP extension is missing.

What it restates (reference file:line, relative to /root/reference):
  * modified ResNet-50                 experiments/siammask_sharp/resnet.py:59-103,151-227
  * ResDownS / ResDown                 experiments/siammask_sharp/custom.py:12-25,58-66
  * DepthCorr + conv2d_dw_group        models/rpn.py:32-72
  * UP / MaskCorr                      experiments/siammask_sharp/custom.py:69-96
  * Refine.forward(test=True)          experiments/siammask_sharp/custom.py:131-154
  * Custom.template/track/track_mask/track_refine
                                       experiments/siammask_sharp/custom.py:173-190
                                       experiments/siammask_base/custom.py:100-112
                                       experiments/siamrpn_resnet/custom.py:87-93
The arithmetic itself lives in PyTorch (torch==0.4.1 pinned at requirements.txt:6), a
third-party dependency that is not under /root/reference; the published semantics of
nn.Conv2d / BatchNorm2d(eval) / MaxPool2d / ConvTranspose2d / F.pad / F.upsample(nearest)
are restated below in plain numpy (im2col + matmul, default float64).

Parity pinning: the reference has NO tests or golden vectors for this path (SURVEY.md
section 4), so this oracle is pinned against outputs of the reference itself, run in this
container on CPU in float64 (oracle/make_golden.py -> tests/golden/*.npz); the check is
tests/test_oracle_golden.py.

``QuantOracle`` additionally emulates the rounding points of the fp16 HIP path (fp16
weights/activations, fp32 accumulate) so that the fp16 kernels can be gated tightly.
"""
import numpy as np

BN_EPS = 1e-5  # nn.BatchNorm2d default, used everywhere in the reference


# ------------------------------------------------------------------------------------
# primitive ops (published torch semantics)
# ------------------------------------------------------------------------------------
def conv2d(x, w, b=None, stride=1, pad=0, dil=1):
    """nn.Conv2d forward, NCHW, cross-correlation (no kernel flip), zero padding."""
    B, C, H, W = x.shape
    Co, Ci, kh, kw = w.shape
    assert Ci == C
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    Hp, Wp = x.shape[2], x.shape[3]
    Ho = (Hp - dil * (kh - 1) - 1) // stride + 1
    Wo = (Wp - dil * (kw - 1) - 1) // stride + 1
    sB, sC, sH, sW = x.strides
    cols = np.lib.stride_tricks.as_strided(
        x, shape=(B, C, kh, kw, Ho, Wo),
        strides=(sB, sC, sH * dil, sW * dil, sH * stride, sW * stride), writeable=False)
    cols = cols.reshape(B, C * kh * kw, Ho * Wo)
    out = np.matmul(w.reshape(Co, -1), cols).reshape(B, Co, Ho, Wo)
    if b is not None:
        out = out + b.reshape(1, -1, 1, 1)
    return out


def conv2d_f32acc(x, w, b=None, stride=1, pad=0, dil=1, kstep=16, ktile=64, ksplit=1):
    """conv2d with the ACCUMULATION MODEL of the MFMA kernels instead of exact sums: K ordered (kh, kw, cin) as the packed
    weights are (smk_kernels.h), every k-step of `kstep` elements is one matrix instruction (products and the in-step sum
    taken as exact, the result rounded once into the float32 accumulator), K tiles of `ktile` elements are dealt round-robin
    to `ksplit` accumulators per k-step position (the WK consumer waves of a workgroup each take every WK-th k-step of a K
    tile) that are added in float32 at the end, then bias.  Inputs / outputs float64 arrays holding fp16-representable
    values, as in QuantOracle.  This is ONE plausible order; the kernels use several (tile shape and K split differ per
    layer and batch size) -- the point of the model is to measure how much the order alone moves the outputs."""
    B, C, H, W = x.shape
    Co, Ci, kh, kw = w.shape
    assert Ci == C
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    Hp, Wp = x.shape[2], x.shape[3]
    Ho = (Hp - dil * (kh - 1) - 1) // stride + 1
    Wo = (Wp - dil * (kw - 1) - 1) // stride + 1
    sB, sC, sH, sW = x.strides
    cols = np.lib.stride_tricks.as_strided(
        x, shape=(B, kh, kw, C, Ho, Wo),
        strides=(sB, sH * dil, sW * dil, sC, sH * stride, sW * stride), writeable=False)
    cols = np.ascontiguousarray(cols.reshape(B, kh * kw * C, Ho * Wo))
    wk = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(Co, kh * kw * C))
    K = kh * kw * C
    acc = [np.zeros((B, Co, Ho * Wo), dtype=np.float32) for _ in range(ksplit)]
    steps_per_tile = max(1, ktile // kstep)
    for i, k0 in enumerate(range(0, K, kstep)):
        part = np.matmul(wk[:, k0:k0 + kstep], cols[:, k0:k0 + kstep, :])            # float64: exact enough for 16 terms
        a = acc[(i % steps_per_tile) % ksplit]
        a += part.astype(np.float32)                                                # one float32 rounding per k-step
    out = acc[0]
    for a in acc[1:]:
        out = out + a
    out = out.astype(np.float64).reshape(B, Co, Ho, Wo)
    if b is not None:
        out = (out.astype(np.float32) + b.reshape(1, -1, 1, 1).astype(np.float32)).astype(np.float64)
    return out


def batchnorm_eval(x, gamma, beta, mean, var):
    """nn.BatchNorm2d in eval(): y = (x - mean) / sqrt(var + eps) * gamma + beta."""
    inv = gamma / np.sqrt(var + BN_EPS)
    return x * inv.reshape(1, -1, 1, 1) + (beta - mean * inv).reshape(1, -1, 1, 1)


def relu(x):
    return np.maximum(x, 0)


def maxpool_3x3_s2_p1(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (pads with -inf)."""
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=-np.inf)
    Ho = (H + 2 - 3) // 2 + 1
    Wo = (W + 2 - 3) // 2 + 1
    sB, sC, sH, sW = xp.strides
    win = np.lib.stride_tricks.as_strided(
        xp, shape=(B, C, Ho, Wo, 3, 3), strides=(sB, sC, 2 * sH, 2 * sW, sH, sW), writeable=False)
    return win.max(axis=(4, 5))


def upsample_nearest(x, size):
    """F.upsample(x, size=(s, s)) with the default mode='nearest':
    src = min(floor(dst * in / out), in - 1)."""
    H, W = x.shape[2], x.shape[3]
    iy = np.minimum((np.arange(size) * H) // size, H - 1)
    ix = np.minimum((np.arange(size) * W) // size, W - 1)
    return x[:, :, iy][:, :, :, ix]


def conv_transpose_1x1_input(x, w, b):
    """nn.ConvTranspose2d(Cin, Cout, k, stride=k) applied to a 1x1 spatial input:
    out[b,o,u,v] = sum_i x[b,i] * w[i,o,u,v] + bias[o]."""
    B = x.shape[0]
    out = np.einsum("bi,iouv->bouv", x.reshape(B, -1), w)
    return out + b.reshape(1, -1, 1, 1)


def conv2d_dw_group(x, kernel):
    """models/rpn.py:32-38 — depth-wise valid cross-correlation per (batch, channel):
    out[b,c,i,j] = sum_{u,v} x[b,c,i+u,j+v] * kernel[b,c,u,v]."""
    B, C, H, W = x.shape
    kb, kc, kh, kw = kernel.shape
    assert (kb, kc) == (B, C), "reference requires template batch == search batch"
    Ho, Wo = H - kh + 1, W - kw + 1
    out = np.zeros((B, C, Ho, Wo), dtype=np.result_type(x, kernel))
    for u in range(kh):
        for v in range(kw):
            out += x[:, :, u:u + Ho, v:v + Wo] * kernel[:, :, u:u + 1, v:v + 1]
    return out


# ------------------------------------------------------------------------------------
# network
# ------------------------------------------------------------------------------------
class Oracle(object):
    """Functional restatement of ``Custom`` for variant in {'rpn','base','sharp'}.

    ``sd`` maps reference state-dict names to numpy arrays.  State carried between calls
    mirrors the reference instance attributes (zf, feature, search, corr_feature)."""

    def __init__(self, sd, variant="sharp", dtype=np.float64):
        self.variant = variant
        self.dtype = dtype
        self.sd = {k: np.asarray(v).astype(dtype) for k, v in sd.items()
                   if not k.endswith("num_batches_tracked")}
        self.zf = None
        self.feature = None
        self.search = None
        self.corr_feature = None
        self.dbg = {}          # intermediates by short name (zk_cls, xs_cls, corr_cls, head0_cls, ...)

    # -- helpers ---------------------------------------------------------------------
    def _conv(self, x, name, stride=1, pad=0, dil=1):
        return conv2d(x, self.sd[name + ".weight"], self.sd.get(name + ".bias"), stride, pad, dil)

    def _bn(self, x, name):
        s = self.sd
        return batchnorm_eval(x, s[name + ".weight"], s[name + ".bias"],
                              s[name + ".running_mean"], s[name + ".running_var"])

    # -- ResNet (experiments/siammask_sharp/resnet.py) ----------------------------------
    def _bottleneck(self, x, p, stride, dil, ds):
        """Bottleneck.forward, resnet.py:80-103; conv2 padding per :66-72;
        ds = None | (kernel, stride, pad) of the shortcut conv (_make_layer :184-206)."""
        pad2 = dil if dil > 1 else 2 - stride
        out = relu(self._bn(self._conv(x, p + "conv1"), p + "bn1"))
        out = relu(self._bn(self._conv(out, p + "conv2", stride, pad2, dil), p + "bn2"))
        out = self._bn(self._conv(out, p + "conv3"), p + "bn3")
        if ds is not None:
            k, s, pd = ds
            residual = self._bn(self._conv(x, p + "downsample.0", s, pd, 1), p + "downsample.1")
        else:
            residual = x
        return relu(out + residual)

    def resnet(self, x):
        """ResNet.forward, resnet.py:217-227 -> (p0, p1, p2, p3)."""
        f = "features.features."
        p0 = relu(self._bn(self._conv(x, f + "conv1", 2, 0), f + "bn1"))      # 7x7 s2 p0 (:154)
        x = maxpool_3x3_s2_p1(p0)                                             # (:158)
        # layer1: stride 1, dilation 1, 1x1 shortcut (:159, :188-193)
        for b in range(3):
            x = self._bottleneck(x, f + "layer1.%d." % b, 1, 1, (1, 1, 0) if b == 0 else None)
        p1 = x
        # layer2: stride 2 -> 3x3 s2 p0 shortcut, conv2 3x3 s2 p0 (:160, :195-206)
        for b in range(4):
            x = self._bottleneck(x, f + "layer2.%d." % b, 2 if b == 0 else 1, 1,
                                 (3, 2, 0) if b == 0 else None)
        p2 = x
        # layer3: stride 1 dilation 2 -> block 0 dilation 1 with 3x3 s1 p1 shortcut,
        # blocks 1..5 dilation 2 (:165, :196-213)
        for b in range(6):
            x = self._bottleneck(x, f + "layer3.%d." % b, 1, 1 if b == 0 else 2,
                                 (3, 1, 1) if b == 0 else None)
        p3 = x
        return p0, p1, p2, p3

    def resdown(self, x):
        """ResDown.forward_all + ResDownS.forward, custom.py:19-25,58-66."""
        feats = self.resnet(x)
        d = "features.downsample.downsample."
        y = self._bn(self._conv(feats[3], d + "0"), d + "1")
        if y.shape[3] < 20:                       # custom.py:21-24 (template only: 15 < 20)
            y = y[:, :, 4:-4, 4:-4]
        return feats, y

    # -- DepthCorr (models/rpn.py:41-72) --------------------------------------------------
    def _conv_bn_relu(self, x, p):
        return relu(self._bn(self._conv(x, p + ".0"), p + ".1"))

    def forward_corr(self, p, kernel, search):
        br = p.rstrip(".").split(".")[-1]
        k = self._conv_bn_relu(kernel, p + "conv_kernel")       # rpn.py:64
        s = self._conv_bn_relu(search, p + "conv_search")       # rpn.py:65
        corr = conv2d_dw_group(s, k)                            # rpn.py:66
        self.dbg["zk_" + br], self.dbg["xs_" + br], self.dbg["corr_" + br] = k, s, corr
        return corr

    def head(self, p, feature):
        h = self._conv_bn_relu(feature, p + "head")             # rpn.py:56-59
        self.dbg["head0_" + p.rstrip(".").split(".")[-1]] = h
        return self._conv(h, p + "head.3")                      # rpn.py:60 (1x1 with bias)

    def depthcorr(self, p, kernel, search):
        return self.head(p, self.forward_corr(p, kernel, search))   # rpn.py:69-72

    # -- Refine (custom.py:131-154, test=True path) ------------------------------------
    def _seq2(self, x, p):
        """nn.Sequential(conv3x3 p1, ReLU, conv3x3 p1, ReLU) of custom.py:102-118."""
        x = relu(self._conv(x, p + ".0", 1, 1))
        return relu(self._conv(x, p + ".2", 1, 1))

    def refine(self, f, corr_feature, pos):
        r = "refine_model."
        y, x = int(pos[0]), int(pos[1])
        pz = lambda t, n: np.pad(t, ((0, 0), (0, 0), (n, n), (n, n)))
        p0 = pz(f[0], 16)[:, :, 4 * y:4 * y + 61, 4 * x:4 * x + 61]        # :133
        p1 = pz(f[1], 8)[:, :, 2 * y:2 * y + 31, 2 * x:2 * x + 31]         # :134
        p2 = pz(f[2], 4)[:, :, y:y + 15, x:x + 15]                         # :135
        p3 = corr_feature[:, :, y, x].reshape(-1, 256, 1, 1)                # :145
        out = conv_transpose_1x1_input(p3, self.sd[r + "deconv.weight"], self.sd[r + "deconv.bias"])  # :149
        out = self._conv(upsample_nearest(self._seq2(out, r + "h2") + self._seq2(p2, r + "v2"), 31),
                         r + "post0", 1, 1)                                 # :150
        out = self._conv(upsample_nearest(self._seq2(out, r + "h1") + self._seq2(p1, r + "v1"), 61),
                         r + "post1", 1, 1)                                 # :151
        out = self._conv(upsample_nearest(self._seq2(out, r + "h0") + self._seq2(p0, r + "v0"), 127),
                         r + "post2", 1, 1)                                 # :152
        return out.reshape(-1, 127 * 127)                                   # :153

    # -- Custom surface ----------------------------------------------------------------
    def template(self, z):
        """custom.py:173-174."""
        _, self.zf = self.resdown(np.asarray(z, dtype=self.dtype))

    def track(self, x):
        """custom.py:176-179 -> (cls [B,10,25,25], loc [B,20,25,25])."""
        _, search = self.resdown(np.asarray(x, dtype=self.dtype))
        cls = self.depthcorr("rpn_model.cls.", self.zf, search)
        loc = self.depthcorr("rpn_model.loc.", self.zf, search)
        return cls, loc

    def track_mask(self, x):
        """sharp: custom.py:181-186; base: experiments/siammask_base/custom.py:108-112."""
        assert self.variant in ("base", "sharp")
        self.feature, self.search = self.resdown(np.asarray(x, dtype=self.dtype))
        cls = self.depthcorr("rpn_model.cls.", self.zf, self.search)
        loc = self.depthcorr("rpn_model.loc.", self.zf, self.search)
        self.corr_feature = self.forward_corr("mask_model.mask.", self.zf, self.search)
        pred_mask = self.head("mask_model.mask.", self.corr_feature)
        return cls, loc, pred_mask

    def track_refine(self, pos):
        """custom.py:188-190.  ``pos`` = (y, x) shared by the batch (reference semantics) or a
        [B,2] array of per-item positions (batched-stream extension; equals running the
        reference once per item)."""
        assert self.variant == "sharp"
        pos = np.asarray(pos)
        if pos.ndim == 1:
            return self.refine(self.feature, self.corr_feature, pos)
        outs = []
        for b in range(pos.shape[0]):
            f = [t[b:b + 1] for t in self.feature]
            outs.append(self.refine(f, self.corr_feature[b:b + 1], pos[b]))
        return np.concatenate(outs, axis=0)


# ------------------------------------------------------------------------------------
# quantisation-aware oracle for the fp16 HIP path ("tight" fp16 gate)
# ------------------------------------------------------------------------------------
def q16(a):
    """round to fp16 the way the device does (f64 -> f32 -> f16), keep computing in float64"""
    return np.asarray(a).astype(np.float32).astype(np.float16).astype(np.float64)


class QuantOracle(Oracle):
    """The reference op sequence with the rounding points of the fp16 kernels:
      * BatchNorm folded into the conv in float64, folded weights rounded to fp16, folded
        bias kept in fp32 (engine.cpp: fold/pack_rows/upload_packed);
      * every activation that the kernels STORE (NHWC, fp16) is rounded to fp16 after the fused
        epilogue (bias, residual, ReLU); accumulation is exact here (fp32 on the device);
      * tensors handed back to the caller (cls, loc, mask, refine logits) are fp32: not rounded.
    Same call surface as ``Oracle``."""

    def __init__(self, sd, variant="sharp", refine_sum_in_h=True, accum="exact", ksplit=1):
        super(QuantOracle, self).__init__(sd, variant, np.float64)
        # which of Refine's two branch outputs is stored (rounded) before the sum: True = v*.2 (the default
        # device path, refine_chain.hip), False = h*.2 (the per-layer path, SMK_TUNE=chain=0)
        self.refine_sum_in_h = refine_sum_in_h
        # accumulation model of the convolutions: "exact" (float64 sums, the default) or "f32" (conv2d_f32acc: float32
        # accumulator, one rounding per 16-element k-step in the device's K order, `ksplit` interleaved accumulators)
        assert accum in ("exact", "f32")
        self.accum, self.ksplit = accum, ksplit

    def _fold(self, conv, bn=None):
        w = self.sd[conv + ".weight"]
        co = w.shape[0]
        scale, shift = np.ones(co), np.zeros(co)
        if bn is not None:
            s = self.sd
            scale = s[bn + ".weight"] / np.sqrt(s[bn + ".running_var"] + BN_EPS)
            shift = s[bn + ".bias"] - s[bn + ".running_mean"] * scale
        if conv + ".bias" in self.sd:
            shift = shift + self.sd[conv + ".bias"]
        return q16(w * scale.reshape(-1, 1, 1, 1)), shift.astype(np.float32).astype(np.float64)

    def _fused(self, x, conv, bn=None, stride=1, pad=0, dil=1, act=False, res=None, res_after_relu=False,
               store=True):
        w, b = self._fold(conv, bn)
        if self.accum == "f32":
            y = conv2d_f32acc(x, w, b, stride, pad, dil, ksplit=self.ksplit)
        else:
            y = conv2d(x, w, b, stride, pad, dil)
        if res is not None and not res_after_relu:
            y = y + res
        if act:
            y = relu(y)
        if res is not None and res_after_relu:
            y = y + res
        return q16(y) if store else y

    def _bottleneck(self, x, p, stride, dil, ds):
        pad2 = dil if dil > 1 else 2 - stride
        out = self._fused(x, p + "conv1", p + "bn1", act=True)
        out = self._fused(out, p + "conv2", p + "bn2", stride, pad2, dil, act=True)
        if ds is not None:
            k, s, pd = ds
            residual = self._fused(x, p + "downsample.0", p + "downsample.1", s, pd, 1)
        else:
            residual = x
        return self._fused(out, p + "conv3", p + "bn3", act=True, res=residual)

    def resnet(self, x):
        f = "features.features."
        x = q16(x)                                                            # cvt_in
        p0 = self._fused(x, f + "conv1", f + "bn1", 2, 0, act=True)
        x = maxpool_3x3_s2_p1(p0)
        for b in range(3):
            x = self._bottleneck(x, f + "layer1.%d." % b, 1, 1, (1, 1, 0) if b == 0 else None)
        p1 = x
        for b in range(4):
            x = self._bottleneck(x, f + "layer2.%d." % b, 2 if b == 0 else 1, 1, (3, 2, 0) if b == 0 else None)
        p2 = x
        for b in range(6):
            x = self._bottleneck(x, f + "layer3.%d." % b, 1, 1 if b == 0 else 2, (3, 1, 1) if b == 0 else None)
        return p0, p1, p2, x

    def resdown(self, x):
        feats = self.resnet(x)
        d = "features.downsample.downsample."
        y = self._fused(feats[3], d + "0", d + "1")
        if y.shape[3] < 20:
            y = y[:, :, 4:-4, 4:-4]
        return feats, y

    def forward_corr(self, p, kernel, search):
        br = p.rstrip(".").split(".")[-1]
        k = self._fused(kernel, p + "conv_kernel.0", p + "conv_kernel.1", act=True)
        s = self._fused(search, p + "conv_search.0", p + "conv_search.1", act=True)
        corr = q16(conv2d_dw_group(s, k))
        self.dbg["zk_" + br], self.dbg["xs_" + br], self.dbg["corr_" + br] = k, s, corr
        return corr

    def head(self, p, feature):
        h = self._fused(feature, p + "head.0", p + "head.1", act=True)
        self.dbg["head0_" + p.rstrip(".").split(".")[-1]] = h
        return self._fused(h, p + "head.3", store=False)

    def refine(self, f, corr_feature, pos):
        r = "refine_model."
        y, x = int(pos[0]), int(pos[1])
        pz = lambda t, n: np.pad(t, ((0, 0), (0, 0), (n, n), (n, n)))
        p0 = pz(f[0], 16)[:, :, 4 * y:4 * y + 61, 4 * x:4 * x + 61]
        p1 = pz(f[1], 8)[:, :, 2 * y:2 * y + 31, 2 * x:2 * x + 31]
        p2 = pz(f[2], 4)[:, :, y:y + 15, x:x + 15]
        p3 = corr_feature[:, :, y, x].reshape(-1, 256, 1, 1)
        out = q16(conv_transpose_1x1_input(p3, q16(self.sd[r + "deconv.weight"]),
                                           self.sd[r + "deconv.bias"].astype(np.float32).astype(np.float64)))

        def stage(out, pf, h, v, post, size, store):
            ha = self._fused(out, r + h + ".0", pad=1, act=True)
            hb = self._fused(ha, r + h + ".2", pad=1, act=True)
            va = self._fused(pf, r + v + ".0", pad=1, act=True)
            if self.refine_sum_in_h:      # device default (refine_chain.hip): v*.2 stored, added after h*.2's ReLU
                vb = self._fused(va, r + v + ".2", pad=1, act=True)
                hb = self._fused(ha, r + h + ".2", pad=1, act=True, res=vb, res_after_relu=True)
                return self._fused(upsample_nearest(hb, size), r + post, pad=1, store=store)
            s = self._fused(va, r + v + ".2", pad=1, act=True, res=hb, res_after_relu=True)
            return self._fused(upsample_nearest(s, size), r + post, pad=1, store=store)

        out = stage(out, p2, "h2", "v2", "post0", 31, True)
        out = stage(out, p1, "h1", "v1", "post1", 61, True)
        out = stage(out, p0, "h0", "v0", "post2", 127, False)
        return out.reshape(-1, 127 * 127)


# ------------------------------------------------------------------------------------
# host-side decode used by the parity tests (tools/test.py:205-254), restated
# ------------------------------------------------------------------------------------
def generate_anchor(score_size=25, stride=8, ratios=(0.33, 0.5, 1, 2, 3), scales=(8,)):
    """tools/test.py:113-129 + utils/anchors.py:28-51 (integer-truncated ws/hs)."""
    anchors = []
    size = stride * stride
    for r in ratios:
        ws = int(np.sqrt(size * 1.0 / r))
        hs = int(ws * r)
        for s in scales:
            anchors.append((ws * s, hs * s))
    anchor_num = len(anchors)
    ori = -(score_size // 2) * stride
    xx, yy = np.meshgrid([ori + stride * dx for dx in range(score_size)],
                         [ori + stride * dy for dy in range(score_size)])
    out = np.zeros((anchor_num * score_size * score_size, 4), dtype=np.float32)
    n = score_size * score_size
    for a, (w, h) in enumerate(anchors):
        out[a * n:(a + 1) * n, 0] = xx.flatten()
        out[a * n:(a + 1) * n, 1] = yy.flatten()
        out[a * n:(a + 1) * n, 2] = w
        out[a * n:(a + 1) * n, 3] = h
    return out


def decode_best(cls, loc, target_sz=(60.0, 80.0), scale_x=1.0, penalty_k=0.04,
                window_influence=0.4, score_size=25, lr=1.0):
    """Per-item restatement of tools/test.py:205-254 WITH THE TOOL'S OWN DTYPES: the arithmetic follows the dtype of
    the network outputs exactly as the NumPy (>= 2) / torch expressions of the tool do.  For the float32 tensors the
    tools see this means: float32 softmax (torch, :206), float32 anchor decode incl. np.exp (:209-212; anchors are
    float32, utils/anchors.py:29), float32 sz() and w/h ratio (:217-220,231-232), and promotion to float64 where a
    float32 array meets an np.float64 scalar -- the division by sz_wh(target_sz_in_crop) / the ratio of the float64
    target size (:231-232) -- so penalty, pscore and the window blend are float64 (:234-237).  float64 inputs (the
    reference run as model.double()) stay float64 throughout.  Pinned bit-for-bit against the unchanged tool by
    tests/test_decode_reference.py (fixtures from oracle/make_tracker_golden.py).
    cls: [10,25,25], loc: [20,25,25] (one item) -> (best_id, delta_y, delta_x, pscore)."""
    import torch
    cls, loc = np.asarray(cls), np.asarray(loc)
    anchor = generate_anchor(score_size)                                   # float32 [3125,4] (cx, cy, w, h)
    delta = loc.reshape(4, -1).copy()                                      # :205
    sc = torch.from_numpy(np.ascontiguousarray(cls.reshape(2, -1).T))      # :206 view(2,-1).permute(1,0)
    score = torch.softmax(sc, dim=1)[:, 1].numpy()
    delta[0, :] = delta[0, :] * anchor[:, 2] + anchor[:, 0]                # :209-212
    delta[1, :] = delta[1, :] * anchor[:, 3] + anchor[:, 1]
    delta[2, :] = np.exp(delta[2, :]) * anchor[:, 2]
    delta[3, :] = np.exp(delta[3, :]) * anchor[:, 3]

    def change(r):
        return np.maximum(r, 1. / r)

    def sz(w, h):
        pad = (w + h) * 0.5
        sz2 = (w + pad) * (h + pad)
        return np.sqrt(sz2)

    # np.float64 scalars, as in the tool (target_sz is a float64 array, scale_x an np.float64): NOT Python floats,
    # which NumPy 2 would treat as weak and keep the float32 arrays float32
    scale_x = np.float64(scale_x)
    target_sz_in_crop = np.asarray(target_sz, dtype=np.float64) * scale_x   # :230
    s_c = change(sz(delta[2, :], delta[3, :]) / sz(target_sz_in_crop[0], target_sz_in_crop[1]))     # :231
    r_c = change((target_sz_in_crop[0] / target_sz_in_crop[1]) / (delta[2, :] / delta[3, :]))       # :232
    penalty = np.exp(-(r_c * s_c - 1) * penalty_k)                          # :234
    pscore = penalty * score                                                # :235
    window = np.outer(np.hanning(score_size), np.hanning(score_size))      # :158-162
    window = np.tile(window.flatten(), 5)
    pscore = pscore * (1 - window_influence) + window * window_influence    # :238
    best = int(np.argmax(pscore))                                           # :239
    _, dy, dx = np.unravel_index(best, (5, score_size, score_size))         # :253-254
    pred_in_crop = delta[:, best] / scale_x                                 # :241
    decode_best.last = {"box": np.array([delta[0, best], delta[1, best], delta[2, best], delta[3, best],
                                          score[best], penalty[best], pscore[best], best], dtype=np.float64),
                        "pred_in_crop": np.asarray(pred_in_crop, dtype=np.float64),
                        "lr": float(penalty[best] * score[best] * lr),      # :242
                        "dtypes": (str(delta.dtype), str(score.dtype), str(penalty.dtype), str(pscore.dtype))}
    return best, int(dy), int(dx), pscore
