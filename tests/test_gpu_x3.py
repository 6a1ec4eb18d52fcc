"""Split-operand fp16 contexts (dtype "f16x3", SMK_DTYPE_F16X3, ABI 1.6) -- north_star: "bit-exact for the argmax box index"
(/root/reference/tools/test.py:237 np.argmax(pscore)) at more than the fp32 matrix pipe's speed.

Every value of the track path's trunk is an fp16 pair hi + lo stored as two channel planes [hi | lo]; a convolution's gather presents them
as the operand [hi | hi | lo] and the weights are packed [w_hi | w_lo | w_hi] per tap, so the fp16 implicit-GEMM kernel on the tripled K forms
x_hi w_hi + x_hi w_lo + x_lo w_hi in its fp32 accumulators (conv_igemm.hip's epilogue splits the result again).  The CPU model of exactly this arithmetic reproduces the fp64
oracle's index on 1024 / 1024 streams (tools/measure/cpu_split_operand_study.py, profiles/r06_cpu_split_operand_study.json); here the
DEVICE is held to it:
  * one convolution (every geometry class of the path) against the float64 oracle at the fp32 context's gate (2e-5);
  * the golden B = 2 inputs end to end: cls / loc against the float64 oracle at the fp32 gate of tests/test_gpu_e2e.py (1e-4), the kept
    trunk tensors p0 .. p3 / search, the device-decoded index = the oracle's; mask / Refine (plain fp16 on the hi planes) at the fp16 gate;
  * the 1024 streams of tests/golden/argmax_oracle_1024.npz: the oracle's index on every one, float32-grade near-ties excepted.
"""
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle.np_oracle import Oracle, decode_best
from helpers import rel_err
from siammask_amd import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

X3_CONVS = [
    # cin, cout, k, stride, pad, dil, hw, B, with_res
    (3, 64, 7, 2, 0, 1, 63, 2, False),        # stem: 3 -> 8 channels, operand [hi | hi | lo] = 24 per tap (taps straddle K tiles)
    (64, 64, 1, 1, 0, 1, 31, 2, False),       # layer1 1x1
    (64, 64, 3, 1, 1, 1, 31, 2, False),       # layer1 3x3
    (256, 1024, 1, 1, 0, 1, 15, 1, True),     # Bottleneck conv3 + residual + ReLU (the residual is a split tensor too)
    (1024, 256, 1, 1, 0, 1, 15, 1, False),    # conv1, K = 3 x 1024
    (256, 256, 3, 1, 2, 2, 15, 1, False),     # layer3 conv2, dilation 2
    (128, 128, 3, 2, 0, 1, 31, 1, False),     # 3x3 stride 2 pad 0
    (256, 10, 1, 1, 0, 1, 25, 2, False),      # N = 10
]


@pytest.mark.parametrize("cfg", X3_CONVS)
def test_split_operand_convolution(cfg):
    from siammask_amd import ops
    cin, cout, k, stride, pad, dil, hw, B, with_res = cfg
    rng = np.random.default_rng(hash(cfg) & 0xffff)
    x = rng.uniform(-1, 1, size=(B, cin, hw, hw)).astype(np.float32)
    w = (rng.uniform(-1, 1, size=(cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.uniform(-1, 1, size=cout).astype(np.float32)
    ho = (hw + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ref = O.conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), stride, pad, dil)
    rd = None
    if with_res:
        res = rng.uniform(-1, 1, size=(B, cout, ho, ho)).astype(np.float32)
        ref = ref + res
        rd = torch.from_numpy(res).cuda()
    ref = np.maximum(ref, 0)
    y = ops.conv2d(torch.from_numpy(x).cuda(), w, b, stride, pad, dil, relu=True, res=rd, res_mode=1, dtype="f16x3")
    e = rel_err(y.cpu().numpy(), ref)
    # fp16 on the same values, for scale: the split operands must buy three orders of magnitude
    y16 = ops.conv2d(torch.from_numpy(x).cuda(), w, b, stride, pad, dil, relu=True, res=rd, res_mode=1, dtype="f16")
    e16 = rel_err(y16.cpu().numpy(), ref)
    print("x3 conv %s: %.2e (plain fp16 %.2e)" % (cfg, e, e16))
    assert e <= 2e-5, (cfg, e, e16)
    # the register-fed kernel's epilogue splits too (wreg_tile.inc): same gate, every tile shape it is launched with in these contexts
    for tile in ((64, 64), (64, 256), (128, 256), (128, 128)):
        yw = ops.conv2d(torch.from_numpy(x).cuda(), w, b, stride, pad, dil, relu=True, res=rd, res_mode=1, dtype="f16x3", algo="wreg", tile=tile)
        ew = rel_err(yw.cpu().numpy(), ref)
        assert ew <= 2e-5, (cfg, tile, ew)


def _model(dtype, B, graph=True):
    from siammask_amd.custom import build
    m = build("sharp", dtype=dtype, graph=graph, max_batch=B)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    return m.eval().cuda()


@pytest.mark.parametrize("B,inputs,graph", [(2, "smooth", True), (8, "random", True), (3, "smooth", False)])
def test_split_operand_context_end_to_end(B, inputs, graph):
    gen = synth.smooth_image_batch if inputs == "smooth" else synth.image_batch
    z = gen(B, 127, stream0=40)
    x = gen(B, 255, stream0=40)
    o = Oracle(synth.state_dict("sharp", "synthetic_damped"), "sharp")
    o.template(z.astype(np.float64))
    ocls, oloc, omask = o.track_mask(x.astype(np.float64))
    twh = np.tile(np.array([[60.0, 80.0]], dtype=np.float64), (B, 1))
    m = _model("f16x3", B, graph)          # (graph = False: eager launches, fresh output tensors per call)
    m.template(torch.from_numpy(z).cuda())
    out = m.track_step(torch.from_numpy(x).cuda(), torch.from_numpy(twh).cuda(), refine=True)
    errs = {"cls": rel_err(out["cls"].cpu().numpy(), ocls), "loc": rel_err(out["loc"].cpu().numpy(), oloc),
            "mask": rel_err(out["mask"].cpu().numpy(), omask)}
    for i, n in enumerate(("p0", "p1", "p2", "p3")):
        errs[n] = rel_err(m.debug_tensor(n).cpu().numpy(), o.feature[i])
    errs["search"] = rel_err(m.debug_tensor("search").cpu().numpy(), o.search)
    box = out["box"].cpu().numpy()
    pos = []
    for b in range(B):
        bid, dy, dx, _ = decode_best(ocls[b], oloc[b], target_sz=(60.0, 80.0), scale_x=1.0)
        pos.append((dy, dx))
        assert int(box[b, 7]) == bid, "stream %d: device argmax %d != oracle %d (%s)" % (b, int(box[b, 7]), bid, errs)
    errs["refine"] = rel_err(out["refine"].cpu().numpy(), o.track_refine(np.asarray(pos)))
    print("f16x3 end to end B=%d %s: %s" % (B, inputs, {k: "%.1e" % v for k, v in errs.items()}))
    exact = {k: v for k, v in errs.items() if k not in ("mask", "refine")}
    bad = {k: v for k, v in exact.items() if not v <= 1e-4}
    assert not bad, "split-operand trunk over the fp32 gate: %s (all %s)" % (bad, errs)
    # mask head and Refine: plain fp16 arithmetic on fp32-grade inputs -> the fp16 context's gate
    assert errs["mask"] <= 5e-3 and errs["refine"] <= 5e-3, errs


@pytest.mark.parametrize("variant", ["rpn", "base"])
def test_split_operand_context_other_variants(variant):
    """the two other model families of the reference (experiments/siamrpn_resnet/custom.py:87-93: two branches, no mask; experiments/siammask_base/
    custom.py:100-112: three branches, the 63 x 63 mask head, no Refine) through the same split-operand trunk: cls / loc at the fp32 gate, the decoded
    index = the oracle's, base's mask at the fp16 gate"""
    from siammask_amd.custom import build
    B = 2
    z = synth.smooth_image_batch(B, 127, stream0=60)
    x = synth.smooth_image_batch(B, 255, stream0=60)
    o = Oracle(synth.state_dict(variant, "synthetic_damped"), variant)
    o.template(z.astype(np.float64))
    m = build(variant, dtype="f16x3", graph=True, max_batch=B)
    m.load_state_dict(synth.torch_state_dict(variant, "synthetic_damped"))
    m = m.eval().cuda()
    m.template(torch.from_numpy(z).cuda())
    twh = np.tile(np.array([[60.0, 80.0]], dtype=np.float64), (B, 1))
    if variant == "rpn":
        ocls, oloc = o.track(x.astype(np.float64))[:2]
        out = m.track_step(torch.from_numpy(x).cuda(), torch.from_numpy(twh).cuda(), refine=False, mask_head=False)
    else:
        ocls, oloc, omask = o.track_mask(x.astype(np.float64))
        out = m.track_step(torch.from_numpy(x).cuda(), torch.from_numpy(twh).cuda(), refine=False)
    errs = {"cls": rel_err(out["cls"].cpu().numpy(), ocls), "loc": rel_err(out["loc"].cpu().numpy(), oloc)}
    if variant == "base":
        errs["mask"] = rel_err(out["mask"].cpu().numpy(), omask)
    box = out["box"].cpu().numpy()
    for b in range(B):
        bid = decode_best(ocls[b], oloc[b], target_sz=(60.0, 80.0), scale_x=1.0)[0]
        assert int(box[b, 7]) == bid, (variant, b, int(box[b, 7]), bid, errs)
    print("f16x3 %s: %s" % (variant, {k: "%.1e" % v for k, v in errs.items()}))
    assert errs["cls"] <= 1e-4 and errs["loc"] <= 1e-4, errs
    assert errs.get("mask", 0.0) <= 5e-3, errs


def test_split_operand_argmax_is_the_oracles_on_1024_streams():
    """the statistic of tests/test_gpu_argmax.py for the f16x3 context: device-decoded best_id vs the fp64 oracle's, all 1024 streams"""
    B, SEEDS = 64, 8
    gold = np.load(os.path.join(REPO, "tests", "golden", "argmax_oracle_1024.npz"))
    kinds = [str(k) for k in gold["kinds"]]
    top_idx, top_ps = gold["top_idx"].astype(np.int64), gold["top_pscore"]
    gen = {"smooth": synth.smooth_image_batch, "noise": synth.image_batch}
    m = _model("f16x3", B)
    exact = near = total = 0
    worst = 0.0
    for ki, kind in enumerate(kinds):
        for seed in range(SEEDS):
            s0 = 10000 * (seed + 1)
            z = torch.from_numpy(gen[kind](B, 127, stream0=s0)).cuda()
            x = torch.from_numpy(gen[kind](B, 255, stream0=s0 + 5000)).cuda()
            g = np.random.Generator(np.random.PCG64(7 + seed))
            twh = torch.from_numpy(g.uniform(40.0, 110.0, size=(B, 2))).cuda()
            m.template(z)
            got = m.track_step(x, twh, refine=False, mask_head=False)["box"].cpu().numpy()[:, 7].astype(np.int64)
            want = top_idx[ki, seed, :, 0]
            same = got == want
            hit = got[:, None] == top_idx[ki, seed]
            ps_pick = np.where(hit.any(-1), (top_ps[ki, seed] * hit).sum(-1), -np.inf)
            deficit = top_ps[ki, seed, :, 0] - ps_pick
            exact += int(same.sum()); total += B
            near += int((~same & (deficit <= 2e-5)).sum())
            worst = max(worst, float(np.where(same, 0.0, np.where(np.isfinite(deficit), deficit, 1.0)).max()))
    print("f16x3 vs the fp64 oracle: %d / %d exact, %d float32-grade near-ties, worst deficit %.3g" % (exact, total, near, worst))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    import json
    with open(os.path.join(REPO, "gpurun_out", "argmax_agreement_f16x3.json"), "w") as f:
        json.dump({"streams": total, "exact": exact, "float_near_ties": near, "worst_deficit_in_oracle_ranking": worst}, f)
    assert total == 1024 and exact + near == 1024 and exact >= 1022, (exact, near, worst)
