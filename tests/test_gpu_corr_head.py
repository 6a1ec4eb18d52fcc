"""corr_head_kernel (round 4): depth-wise cross-correlation + head.0 + cls / loc head.3 as ONE launch (fp16;
/root/reference/models/rpn.py:32-38,50-72) against the three launches it replaces, in one process: the correlation must come
out BIT-identical (same fp32 fmaf order, same fp16 rounding), head.0 and the cls / loc logits within fp16 summation-order noise
of the per-launch kernels -- and both against the quantisation-aware oracle at the tight fp16 gate.  All variants (three branches
for sharp / base, two for rpn), odd batches, and the lazy mask head."""
import numpy as np
import pytest
import torch

from helpers import rel_err
from siammask_amd import synth

pytestmark = pytest.mark.gpu


def _model(variant, B):
    from siammask_amd.custom import build
    m = build(variant, dtype="f16", graph=True, max_batch=B)
    m.load_state_dict(synth.torch_state_dict(variant, "synthetic_damped"))
    return m.eval().cuda()


def _run(variant, B, knob):
    from siammask_amd import _lib
    old = _lib.tune_get("corr_head")
    try:
        _lib.tune(corr_head=knob)
        m = _model(variant, B)
        z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=400)).cuda()
        x = torch.from_numpy(synth.image_batch(B, 255, stream0=400)).cuda()
        m.template(z)
        if variant == "rpn":
            cls, loc = m.track(x)
            mask = None
        else:
            cls, loc, mask = m.track_mask(x)
        out = {"cls": cls.clone(), "loc": loc.clone(), "corr": m.debug_tensor("corr").clone(), "head0": m.debug_tensor("head0").clone()}
        if mask is not None:
            out["mask"] = mask.clone()
        m.profile(2)
        (m.track(x) if variant == "rpn" else m.track_mask(x))
        kernels = [r["kernel"].split("<")[0] for r in m.profile_dump()]
        m.profile(0)
        torch.cuda.synchronize()
        return out, kernels
    finally:
        _lib.tune(corr_head=old)


@pytest.mark.parametrize("variant,B", [("sharp", 8), ("sharp", 1), ("base", 3), ("rpn", 2)])
def test_corr_head_fusion_equals_the_three_launches(variant, B):
    fused, kf = _run(variant, B, 1)
    plain, kp = _run(variant, B, 0)
    assert "corr_head" in kf and "dw_xcorr" not in kf, kf
    assert "dw_xcorr" in kp and "corr_head" not in kp, kp
    assert len(kf) == len(kp) - 2, (kf, kp)                       # three launches became one
    assert torch.equal(fused["corr"], plain["corr"]), "the correlation must be bit-identical (same fmaf order, same rounding)"
    for k in ("head0", "cls", "loc") + (("mask",) if "mask" in fused else ()):
        e = rel_err(fused[k].cpu().numpy(), plain[k].cpu().numpy().astype(np.float64))
        assert e <= 2e-3, "%s B=%d: %s differs from the per-launch kernels by %.2e" % (variant, B, k, e)
    # nothing outside the real channels / pixels was touched: the logits' shapes are the reference's
    assert tuple(fused["cls"].shape) == (B, 10, 25, 25) and tuple(fused["loc"].shape) == (B, 20, 25, 25)
    assert torch.isfinite(fused["cls"]).all() and torch.isfinite(fused["loc"]).all()


def _backbone(B, knob):
    from siammask_amd import _lib
    old = _lib.tune_get("pair_launch")
    try:
        _lib.tune(pair_launch=knob)
        m = _model("sharp", B)
        z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=420)).cuda()
        x = torch.from_numpy(synth.image_batch(B, 255, stream0=420)).cuda()
        m.template(z)
        zf = m.debug_tensor("zf").clone()
        cls, loc, mask = m.track_mask(x)
        out = {"zf": zf, "cls": cls.clone(), "loc": loc.clone()}
        for n in ("p2", "p3", "search"):
            out[n] = m.debug_tensor(n).clone()
        m.profile(2)
        m.track_mask(x)
        kernels = [r["kernel"].split("<")[0] for r in m.profile_dump()]
        m.profile(0)
        torch.cuda.synchronize()
        return out, kernels
    finally:
        _lib.tune(pair_launch=old)


@pytest.mark.parametrize("B", [2, 9])
def test_pair_launch_on_64_row_tiles_equals_two_launches(B):
    """the 64-row form (c3c1s_tile: conv3 in two channel halves, the full Y image in LDS; smk_tune pair_launch = 3) against the two
    launches: ragged last tiles (B x 961 and B x 225 rows are no multiples of 64), both shapes, adjust"""
    fused, kf = _backbone(B, 3)
    plain, kp = _backbone(B, 0)
    assert kf.count("conv_pair") == 9, kf
    for k in ("p2", "p3", "search", "zf", "cls", "loc"):
        e = rel_err(fused[k].cpu().numpy(), plain[k].cpu().numpy().astype(np.float64))
        assert e <= 3e-3, "B=%d: %s differs from the two-launch path by %.2e" % (B, k, e)


@pytest.mark.parametrize("B", [1, 3, 10])
def test_pair_launch_equals_two_launches(B):
    """conv_pair_kernel (round 4): outside the persistent sequence (B = 1, 3, 10 here) every identity Bottleneck's conv3 + the next
    1x1 convolution -- and layer3's last conv3 + adjust -- run as ONE launch (the sequence's c3c1_tile per 32 rows of the flattened
    batch).  Against the two launches: p2 / p3 / search / zf and the logits within fp16 summation-order noise; 9 launches fewer on
    the search branch (3 pairs in layer2, 5 + adjust in layer3)."""
    fused, kf = _backbone(B, 2)                  # (2 = at every batch; the default rule takes 3 <= B <= 31)
    plain, kp = _backbone(B, 0)
    assert kf.count("conv_pair") == 9, kf
    assert "conv_pair" not in kp and len(kp) == len(kf) + 9, (len(kp), len(kf))
    for k in ("p2", "p3", "search", "zf", "cls", "loc"):
        e = rel_err(fused[k].cpu().numpy(), plain[k].cpu().numpy().astype(np.float64))
        assert e <= 3e-3, "B=%d: %s differs from the two-launch path by %.2e" % (B, k, e)


def _refine(B, knob):
    from siammask_amd import _lib
    old = _lib.tune_get("rf_wreg")
    try:
        _lib.tune(rf_wreg=knob)
        m = _model("sharp", B)
        z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=410)).cuda()
        x = torch.from_numpy(synth.image_batch(B, 255, stream0=410)).cuda()
        m.template(z)
        m.track_mask(x)
        pos = [(6 * b) % 25 for b in range(B)]                  # 0 and 24 (windows over the image border) included
        ref = m.track_refine([(p, 24 - p) for p in pos]).clone()
        m.profile(2)
        m.track_refine([(p, 24 - p) for p in pos])
        kernels = [r["kernel"] for r in m.profile_dump()]
        m.profile(0)
        torch.cuda.synchronize()
        return ref, kernels
    finally:
        _lib.tune(rf_wreg=old)


@pytest.mark.parametrize("B", [1, 8, 13])
def test_refine_front_on_the_register_fed_kernel(B):
    """Refine's two merged front launches (v2.0 / v1.0 / v0.0 + deconv, then the three v*.2; /root/reference/experiments/siammask_sharp/
    custom.py:102-118,133-152) run on conv_wreg_kernel's 64x64 tiles by default (smk_tune rf_wreg = 3: hundreds of short-K workgroups,
    where the LDS-staged kernel's per-workgroup set-up is what costs).  Same K split, same fp32 order: the 127 x 127 logits must be
    BIT-identical to the LDS-staged launches', windows at the image border included."""
    new, kn = _refine(B, 3)
    old, ko = _refine(B, 0)
    assert sum("conv_wreg" in k and "merged" in k for k in kn) == 2, kn
    assert sum("conv_igemm" in k and "merged" in k for k in ko) == 2, ko
    assert torch.equal(new, old)
