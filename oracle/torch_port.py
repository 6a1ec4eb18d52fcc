"""CPU baseline port: the same op sequence as the reference's Custom, as plain torch
functional calls on the host cores (fp32).

    *** TEST / BASELINE INFRASTRUCTURE ONLY *** (see oracle/np_oracle.py header)

The reference's CPU path IS this: nn.Conv2d / BatchNorm2d(eval) / ReLU / MaxPool2d /
F.conv2d(groups) / ConvTranspose2d / F.pad / nearest upsample dispatched to ATen
(experiments/siammask_sharp/custom.py:131-190, resnet.py:80-103,217-227, models/rpn.py:32-72).
/root/reference does not exist on the GPU box, so bench.py times this port ("kind": "port")
beside the GPU number.  It is pinned against the golden vectors by tests/test_oracle_golden.py.
"""
import torch
import torch.nn.functional as F


class TorchPort(object):
    def __init__(self, sd, variant="sharp", dtype=torch.float32):
        self.variant = variant
        self.sd = {k: torch.as_tensor(v).to(dtype) for k, v in sd.items() if not k.endswith("num_batches_tracked")}
        self.zf = self.feature = self.search = self.corr_feature = None

    def _cbr(self, x, conv, bn, stride=1, pad=0, dil=1, relu=True):
        s = self.sd
        x = F.conv2d(x, s[conv + ".weight"], None, stride, pad, dil)
        x = F.batch_norm(x, s[bn + ".running_mean"], s[bn + ".running_var"], s[bn + ".weight"], s[bn + ".bias"],
                         False, 0.0, 1e-5)
        return F.relu(x) if relu else x

    def _bottleneck(self, x, p, stride, dil, ds):      # resnet.py:80-103
        pad2 = dil if dil > 1 else 2 - stride
        out = self._cbr(x, p + "conv1", p + "bn1")
        out = self._cbr(out, p + "conv2", p + "bn2", stride, pad2, dil)
        out = self._cbr(out, p + "conv3", p + "bn3", relu=False)
        res = x if ds is None else self._cbr(x, p + "downsample.0", p + "downsample.1", ds[1], ds[2], 1, relu=False)
        return F.relu(out + res)

    def resdown(self, x):                               # resnet.py:217-227 + custom.py:19-25
        f = "features.features."
        p0 = self._cbr(x, f + "conv1", f + "bn1", 2, 0)
        x = F.max_pool2d(p0, 3, 2, 1)
        for b in range(3):
            x = self._bottleneck(x, f + "layer1.%d." % b, 1, 1, (1, 1, 0) if b == 0 else None)
        p1 = x
        for b in range(4):
            x = self._bottleneck(x, f + "layer2.%d." % b, 2 if b == 0 else 1, 1, (3, 2, 0) if b == 0 else None)
        p2 = x
        for b in range(6):
            x = self._bottleneck(x, f + "layer3.%d." % b, 1, 1 if b == 0 else 2, (3, 1, 1) if b == 0 else None)
        d = "features.downsample.downsample."
        y = self._cbr(x, d + "0", d + "1", relu=False)
        if y.size(3) < 20:
            y = y[:, :, 4:-4, 4:-4]
        return (p0, p1, p2, x), y

    def forward_corr(self, p, z, x):                    # models/rpn.py:63-67
        k = self._cbr(z, p + "conv_kernel.0", p + "conv_kernel.1")
        s = self._cbr(x, p + "conv_search.0", p + "conv_search.1")
        B, C = k.shape[:2]
        out = F.conv2d(s.reshape(1, B * C, s.size(2), s.size(3)), k.reshape(B * C, 1, k.size(2), k.size(3)),
                       groups=B * C)
        return out.reshape(B, C, out.size(2), out.size(3))

    def head(self, p, f):                               # models/rpn.py:56-61
        h = self._cbr(f, p + "head.0", p + "head.1")
        return F.conv2d(h, self.sd[p + "head.3.weight"], self.sd[p + "head.3.bias"])

    def template(self, z):
        _, self.zf = self.resdown(z)

    def track(self, x):
        _, s = self.resdown(x)
        return (self.head("rpn_model.cls.", self.forward_corr("rpn_model.cls.", self.zf, s)),
                self.head("rpn_model.loc.", self.forward_corr("rpn_model.loc.", self.zf, s)))

    def track_mask(self, x):
        self.feature, self.search = self.resdown(x)
        cls = self.head("rpn_model.cls.", self.forward_corr("rpn_model.cls.", self.zf, self.search))
        loc = self.head("rpn_model.loc.", self.forward_corr("rpn_model.loc.", self.zf, self.search))
        self.corr_feature = self.forward_corr("mask_model.mask.", self.zf, self.search)
        return cls, loc, self.head("mask_model.mask.", self.corr_feature)

    def _seq2(self, x, p):
        s = self.sd
        x = F.relu(F.conv2d(x, s[p + ".0.weight"], s[p + ".0.bias"], 1, 1))
        return F.relu(F.conv2d(x, s[p + ".2.weight"], s[p + ".2.bias"], 1, 1))

    def track_refine(self, pos):                        # custom.py:131-154
        s, r = self.sd, "refine_model."
        f, y, x = self.feature, int(pos[0]), int(pos[1])
        p0 = F.pad(f[0], [16] * 4)[:, :, 4 * y:4 * y + 61, 4 * x:4 * x + 61]
        p1 = F.pad(f[1], [8] * 4)[:, :, 2 * y:2 * y + 31, 2 * x:2 * x + 31]
        p2 = F.pad(f[2], [4] * 4)[:, :, y:y + 15, x:x + 15]
        p3 = self.corr_feature[:, :, y, x].reshape(-1, 256, 1, 1)
        out = F.conv_transpose2d(p3, s[r + "deconv.weight"], s[r + "deconv.bias"], 15)
        post = lambda t, n: F.conv2d(t, s[r + n + ".weight"], s[r + n + ".bias"], 1, 1)
        out = post(F.interpolate(self._seq2(out, r + "h2") + self._seq2(p2, r + "v2"), size=(31, 31)), "post0")
        out = post(F.interpolate(self._seq2(out, r + "h1") + self._seq2(p1, r + "v1"), size=(61, 61)), "post1")
        out = post(F.interpolate(self._seq2(out, r + "h0") + self._seq2(p0, r + "v0"), size=(127, 127)), "post2")
        return out.reshape(-1, 127 * 127)
