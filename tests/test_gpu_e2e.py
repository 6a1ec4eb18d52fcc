"""End-to-end GPU parity of the drop-in Custom (all arithmetic in libsiammask_hip.so, called
through the C ABI) against (a) the golden vectors produced by the reference itself
(tests/golden, oracle/make_golden.py) and (b) the CPU oracle for every intermediate tensor.

Tolerances, rel. to max|ref| per tensor:
  fp32 path: 1e-4 on synthetic_damped, 5e-4 on synthetic_stress; argmax box index bit-exact.
  fp16 path: 3e-2 on cls/loc/mask, 1e-2 on the refine logits (loose gate vs the fp32 truth;
             the synthetic net amplifies perturbations ~27x); argmax reported.
Per-tensor errors of every run are written to gpurun_out/e2e_*.json."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.np_oracle import Oracle, decode_best
from siammask_amd import synth
from helpers import CASES, load_golden, rel_err

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")


def _model(variant, fixture, dtype, graph, **kw):
    from siammask_amd.custom import build
    m = build(variant, dtype=dtype, graph=graph, **kw)
    m.load_state_dict(synth.torch_state_dict(variant, fixture))
    return m.eval().cuda()


def _cat(o, key, names):
    return np.concatenate([o.dbg["%s_%s" % (key, n)] for n in names], axis=1)


def _run_case(case, dtype, graph):
    g = load_golden(case)
    variant, fixture = str(g["variant"]), str(g["fixture"])
    z = torch.from_numpy(g["z_u8"].astype(np.float32)).cuda()
    x = torch.from_numpy(g["x_u8"].astype(np.float32)).cuda()
    B = x.shape[0]
    m = _model(variant, fixture, dtype, graph)
    o = Oracle(synth.state_dict(variant, fixture), variant)
    o.template(g["z_u8"].astype(np.float64))
    errs = {}
    m.template(z)
    errs["zf"] = rel_err(m.debug_tensor("zf").cpu().numpy(), o.zf)
    names = ["cls", "loc"] + (["mask"] if variant != "rpn" else [])
    if variant == "rpn":
        cls, loc = m.track(x)
        ocls, oloc = o.track(g["x_u8"].astype(np.float64))
        mask = None
    else:
        cls, loc, mask = m.track_mask(x)
        ocls, oloc, omask = o.track_mask(g["x_u8"].astype(np.float64))
    errs["zk"] = rel_err(m.debug_tensor("zk").cpu().numpy(), _cat(o, "zk", names))
    if variant != "rpn":
        for i, n in enumerate(("p0", "p1", "p2")):
            errs[n] = rel_err(m.debug_tensor(n).cpu().numpy(), o.feature[i])
    errs["search"] = rel_err(m.debug_tensor("search").cpu().numpy(), o.search if variant != "rpn" else
                             o.resdown(g["x_u8"].astype(np.float64))[1])
    for key in ("xs", "corr", "head0"):
        errs[key] = rel_err(m.debug_tensor(key).cpu().numpy(), _cat(o, key, names))
    cls_np, loc_np = cls.cpu().numpy(), loc.cpu().numpy()
    errs["cls"] = rel_err(cls_np, g["cls"])
    errs["loc"] = rel_err(loc_np, g["loc"])
    errs["cls_vs_oracle"] = rel_err(cls_np, ocls)
    best_ok = []
    for b in range(B):
        bid, dy, dx, _ = decode_best(cls_np[b], loc_np[b])
        best_ok.append(bid == int(g["best_id"][b]))
    if mask is not None:
        mk = mask.cpu().numpy()
        errs["mask"] = rel_err(mk, omask)
        errs["mask_col"] = max(rel_err(mk[b, :, g["best_yx"][b][0], g["best_yx"][b][1]], g["mask_col"][b])
                               for b in range(B))
    if variant == "sharp":
        r = m.track_refine(g["best_yx"]).cpu().numpy()
        errs["refine"] = rel_err(r, g["refine"])
        iou = []
        for b in range(B):
            a_, b_ = r[b] > 0, g["refine"][b] > 0
            iou.append(float((a_ & b_).sum() / max(1, (a_ | b_).sum())))
        errs["refine_iou_min"] = min(iou)
        sp = tuple(int(v) for v in g["shared_pos"])
        errs["refine_shared"] = rel_err(m.track_refine(sp).cpu().numpy(), g["refine_shared"])
    report = {"case": case, "dtype": dtype, "graph": graph, "errors": errs, "argmax_equal": best_ok,
              "top2_gap": [float(v) for v in g["top2_gap"]]}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "e2e_%s_%s_g%d.json" % (case, dtype, int(graph))), "w") as f:
        json.dump(report, f, indent=1)
    return report, fixture


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_fp32_matches_reference(case, graph):
    rep, fixture = _run_case(case, "f32", graph)
    tol = 5e-4 if fixture == "synthetic_stress" else 1e-4
    bad = {k: v for k, v in rep["errors"].items() if k != "refine_iou_min" and not v <= tol}
    assert not bad, "fp32 %s: %s (all %s)" % (case, bad, rep["errors"])
    assert all(rep["argmax_equal"]), "argmax box index differs: %s" % rep
    if "refine_iou_min" in rep["errors"]:
        assert rep["errors"]["refine_iou_min"] >= 0.999


@pytest.mark.parametrize("case", ["sharp_damped_b2", "base_damped_b1", "rpn_damped_b1"])
def test_fp16_loose_gate(case):
    rep, _ = _run_case(case, "f16", True)
    e = rep["errors"]
    for k in ("cls", "loc", "mask"):
        if k in e:
            assert e[k] <= 3e-2, "fp16 %s %s: %s" % (case, k, e)
    if "refine" in e:
        assert e["refine"] <= 1e-2 and e["refine_iou_min"] >= 0.995, e


@pytest.mark.parametrize("variant", ["sharp", "base"])
def test_fp16_tight_gate_vs_quant_oracle(variant):
    """fp16 path against the quantisation-aware oracle (same rounding points: fp16 folded weights,
    fp16 stored activations, wide accumulation): <= 5e-3 of max|ref| on every returned tensor and
    every kept intermediate; the loose gate above is against the fp32 truth.  SURVEY.md 8c proposed 2e-3;
    measured on MI355X: 3e-4 .. 3.4e-3 (fp32 sequential accumulation on the device vs exact sums in the
    oracle flips a few fp16 roundings, which the network amplifies), so the gate is 5e-3."""
    from oracle.np_oracle import QuantOracle
    fixture = "synthetic_damped"
    z = synth.smooth_image_batch(2, 127, stream0=11)
    x = synth.smooth_image_batch(2, 255, stream0=11)
    q = QuantOracle(synth.state_dict(variant, fixture), variant)
    q.template(z)
    qcls, qloc, qmask = q.track_mask(x)
    m = _model(variant, fixture, "f16", True)
    m.template(torch.from_numpy(z).cuda())
    cls, loc, mask = m.track_mask(torch.from_numpy(x).cuda())
    errs = {"zf": rel_err(m.debug_tensor("zf").cpu().numpy(), q.zf),
            "search": rel_err(m.debug_tensor("search").cpu().numpy(), q.search),
            "corr": rel_err(m.debug_tensor("corr").cpu().numpy(), _cat(q, "corr", ["cls", "loc", "mask"])),
            "cls": rel_err(cls.cpu().numpy(), qcls), "loc": rel_err(loc.cpu().numpy(), qloc),
            "mask": rel_err(mask.cpu().numpy(), qmask)}
    for i, n in enumerate(("p0", "p1", "p2")):
        errs[n] = rel_err(m.debug_tensor(n).cpu().numpy(), q.feature[i])
    if variant == "sharp":
        pos = np.array([[12, 12], [7, 16]], dtype=np.int32)
        errs["refine"] = rel_err(m.track_refine(pos).cpu().numpy(), q.track_refine(pos))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "e2e_fp16_tight_%s.json" % variant), "w") as f:
        json.dump(errs, f, indent=1)
    bad = {k: v for k, v in errs.items() if not v <= 5e-3}
    assert not bad, "fp16 tight gate %s: %s (all %s)" % (variant, bad, errs)


def test_refine_chain_equals_layer_path():
    """fp16: Refine's tail as ONE launch (refine_chain_kernel, activations in LDS) against the same nine
    convolutions as separate launches of the generic kernel, and against the fp64 oracle.  Positions at the
    score-map corners exercise the zero-padded feature windows."""
    from siammask_amd import _lib
    fixture = "synthetic_damped"
    z = synth.smooth_image_batch(3, 127, stream0=21)
    x = synth.smooth_image_batch(3, 255, stream0=21)
    pos = np.array([[0, 0], [24, 24], [12, 7]], dtype=np.int32)
    got = {}
    try:
        for chain in (0, 1):
            _lib.tune(chain=chain)
            m = _model("sharp", fixture, "f16", True)
            m.template(torch.from_numpy(z).cuda())
            m.track_mask(torch.from_numpy(x).cuda())
            got[chain] = m.track_refine(pos).cpu().numpy()
    finally:
        _lib.tune(chain=1)
    o = Oracle(synth.state_dict("sharp", fixture), "sharp")
    o.template(z.astype(np.float64))
    o.track_mask(x.astype(np.float64))
    ref = o.track_refine(pos)
    errs = {"chain_vs_layers": rel_err(got[1], got[0]), "chain_vs_oracle": rel_err(got[1], ref),
            "layers_vs_oracle": rel_err(got[0], ref)}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "e2e_refine_chain.json"), "w") as f:
        json.dump(errs, f, indent=1)
    assert errs["chain_vs_layers"] <= 5e-3, errs          # same rounding points, different summation order
    assert errs["chain_vs_oracle"] <= 1e-2, errs          # the loose fp16 gate of the refine logits


def test_batch_invariance_and_order_errors():
    """B=2 equals two B=1 runs; call-order and batch-mismatch errors surface as exceptions."""
    g = load_golden("sharp_damped_b2")
    z = torch.from_numpy(g["z_u8"].astype(np.float32)).cuda()
    x = torch.from_numpy(g["x_u8"].astype(np.float32)).cuda()
    m = _model("sharp", "synthetic_damped", "f32", False, max_batch=2)
    with pytest.raises(RuntimeError):
        m.track_mask(x)                      # before template
    m.template(z)
    c2, l2, k2 = [t.clone() for t in m.track_mask(x)]
    r2 = m.track_refine(g["best_yx"]).clone()
    with pytest.raises(RuntimeError):
        m.track(x[:1])                       # batch != template batch (models/rpn.py:33)
    for b in range(2):
        m.template(z[b:b + 1])
        c1, l1, k1 = m.track_mask(x[b:b + 1])
        r1 = m.track_refine(tuple(int(v) for v in g["best_yx"][b]))
        for a_, b_ in ((c1, c2[b:b + 1]), (l1, l2[b:b + 1]), (k1, k2[b:b + 1]), (r1, r2[b:b + 1])):
            assert (a_ - b_).abs().max().item() <= 1e-5 * b_.abs().max().item()
    cb, lb = m.track(x[1:2])                 # box-only path equals the mask path's cls/loc
    assert (cb - c2[1:2]).abs().max().item() <= 1e-5 * c2.abs().max().item()
    assert (lb - l2[1:2]).abs().max().item() <= 1e-5 * l2.abs().max().item()


def test_lazy_mask_head_matches():
    g = load_golden("sharp_damped_b2")
    z = torch.from_numpy(g["z_u8"].astype(np.float32)).cuda()
    x = torch.from_numpy(g["x_u8"].astype(np.float32)).cuda()
    m = _model("sharp", "synthetic_damped", "f32", True, lazy_mask=True)
    m.template(z)
    cls, loc, mask = m.track_mask(x)
    assert mask is None
    assert rel_err(cls.cpu().numpy(), g["cls"]) <= 1e-4
    assert rel_err(m.track_refine(g["best_yx"]).cpu().numpy(), g["refine"]) <= 1e-4


def test_device_decode_matches_host_decode():
    """smk_decode vs the oracle's restatement of tools/test.py:205-254: argmax index bit-exact,
    box values within 1e-5 (float64 on both sides, float32 storage)."""
    from siammask_amd.custom import build
    g = load_golden("sharp_damped_b2")
    m = _model("sharp", "synthetic_damped", "f32", False)
    m.set_tracker_hp(0.04, 0.4)
    cls = torch.from_numpy(g["cls"]).cuda()
    loc = torch.from_numpy(g["loc"]).cuda()
    for twh in ((60.0, 80.0), (33.0, 121.5)):
        t = torch.tensor([twh, twh], dtype=torch.float64).cuda()
        pos, box = m.decode(cls, loc, t)
        pos, box = pos.cpu().numpy(), box.cpu().numpy()
        for b in range(2):
            bid, dy, dx, _ = decode_best(g["cls"][b], g["loc"][b], target_sz=twh)
            assert int(box[b, 7]) == bid and tuple(pos[b]) == (dy, dx)
            want = decode_best.last["box"]
            np.testing.assert_allclose(box[b, :7], want[:7], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_fused_step_equals_separate_calls(dtype):
    """track_step (track -> device decode -> refine, one graph, fork/join concurrency) gives the
    same tensors as track_mask + host decode + track_refine."""
    g = load_golden("sharp_damped_b2")
    z = torch.from_numpy(g["z_u8"].astype(np.float32)).cuda()
    x = torch.from_numpy(g["x_u8"].astype(np.float32)).cuda()
    m = _model("sharp", "synthetic_damped", dtype, True)
    m.template(z)
    cls, loc, mask = m.track_mask(x)
    cls, loc, mask = cls.clone(), loc.clone(), mask.clone()
    best = [decode_best(cls[b].cpu().numpy(), loc[b].cpu().numpy()) for b in range(2)]
    ref = m.track_refine(np.array([[t[1], t[2]] for t in best])).clone()
    twh = torch.tensor([[60.0, 80.0]] * 2, dtype=torch.float64).cuda()
    out = m.track_step(x, twh)
    torch.cuda.synchronize()
    assert torch.equal(out["cls"], cls) and torch.equal(out["loc"], loc) and torch.equal(out["mask"], mask)
    assert [int(v) for v in out["box"][:, 7].cpu()] == [t[0] for t in best]
    assert torch.equal(out["refine"], ref)
    if dtype == "f32":
        assert rel_err(out["refine"].cpu().numpy(), g["refine"]) <= 1e-4


def test_packed_weight_cache(tmp_path):
    """packed-weight cache keyed on the checkpoint.  A hit uploads the stored blob and
    gives bit-identical outputs; a changed checkpoint or a damaged blob is a miss (rebuilt); a blob of
    another dtype is rejected by the library."""
    from siammask_amd import _lib
    from siammask_amd.custom import build
    cache = str(tmp_path / "packs")
    sd = synth.torch_state_dict("sharp", "synthetic_damped")
    z = torch.from_numpy(synth.smooth_image_batch(1, 127, stream0=3)).cuda()
    x = torch.from_numpy(synth.smooth_image_batch(1, 255, stream0=3)).cuda()

    def run(m):
        m.template(z)
        cls, loc, mask = m.track_mask(x)
        ref = m.track_refine((12, 11))
        return [t.clone() for t in (cls, loc, mask, ref)]

    def make(dtype="f16", state=sd):
        m = build("sharp", dtype=dtype, pack_cache=cache)
        m.load_state_dict(state)
        return m.eval().cuda()

    m1 = make()
    o1 = run(m1)
    assert m1.pack_cache_hit is False
    files = os.listdir(cache)
    assert len(files) == 1 and files[0].endswith(".smkpack")
    m2 = make()
    o2 = run(m2)
    assert m2.pack_cache_hit is True
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    # a different checkpoint is a different key
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["rpn_model.cls.head.3.bias"] += 0.5
    m3 = make(state=sd2)
    o3 = run(m3)
    assert m3.pack_cache_hit is False and len(os.listdir(cache)) == 2
    assert not torch.equal(o3[0], o1[0])
    # a damaged blob is rebuilt, not trusted
    path = os.path.join(cache, files[0])
    blob = bytearray(open(path, "rb").read())
    open(path, "wb").write(bytes(blob[:len(blob) // 2]))
    m4 = make()
    o4 = run(m4)
    assert m4.pack_cache_hit is False
    for a, b in zip(o1, o4):
        assert torch.equal(a, b)
    # explicit load of a blob packed for another dtype fails loudly
    m5 = build("sharp", dtype="f32")
    with pytest.raises(_lib.SmkError):
        m5.load_packed(path, device="cuda:0")


@pytest.mark.parametrize("inputs", ["smooth", "random"])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_bench_configuration_b8_end_to_end(dtype, inputs):
    """The configuration bench.py measures (sharp + Refine, B=8 streams, fused step graph): at this batch
    the engine picks the large tiles, the merged launches and -- fp16 -- the persistent sequence (conv_seq_kernel:
    layer2, layer3, adjust), which the B<=2 golden cases do not reach.  fp32 against the float64 oracle (<=1e-4,
    argmax position identical per stream), fp16 against the quantisation-aware oracle (<=5e-3), INCLUDING the tensors
    the sequence itself produces (p2 = layer2 output, p3 = layer3 output, search = adjust output) and p0 / p1 in front of
    it.  inputs: the smooth blobs of the other gates, and the white-noise crops bench.py times (synth.image_batch)."""
    from oracle.np_oracle import QuantOracle
    B = 8
    gen = synth.smooth_image_batch if inputs == "smooth" else synth.image_batch
    z = gen(B, 127, stream0=40)
    x = gen(B, 255, stream0=40)
    sd = synth.state_dict("sharp", "synthetic_damped")
    o = Oracle(sd, "sharp") if dtype == "f32" else QuantOracle(sd, "sharp")
    o.template(z.astype(np.float64))
    ocls, oloc, omask = o.track_mask(x.astype(np.float64))
    twh = np.tile(np.array([[60.0, 80.0]], dtype=np.float64), (B, 1))
    m = _model("sharp", "synthetic_damped", dtype, True, max_batch=B)
    m.template(torch.from_numpy(z).cuda())
    out = m.track_step(torch.from_numpy(x).cuda(), torch.from_numpy(twh).cuda(), refine=True)
    # fp16: 5e-3 on the smooth crops (the gate of every other fp16 test); 6e-3 on white noise, the worst-conditioned input
    # there is (no spatial correlation: the synthetic net amplifies a flipped fp16 rounding of layer1 by ~4x up to cls / loc --
    # measured 5.0-5.2e-3 there with p1 at 1.3e-3, p3 at 3.1e-3, search at 4.1e-3; which roundings flip depends on the
    # summation order of whichever kernel a layer runs on)
    tol = 1e-4 if dtype == "f32" else (5e-3 if inputs == "smooth" else 6e-3)
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
        if dtype == "f32":
            assert int(box[b, 7]) == bid, "stream %d: device argmax %d != oracle %d" % (b, int(box[b, 7]), bid)
    if dtype == "f32":
        oref = o.track_refine(np.asarray(pos))
    else:   # refine at the positions the DEVICE decoded (its fp16 argmax may legitimately differ)
        best = box[:, 7].astype(np.int64)
        oref = o.track_refine(np.stack([(best % 625) // 25, best % 25], 1))
    errs["refine"] = rel_err(out["refine"].cpu().numpy(), oref)
    if dtype == "f16":
        assert m.seq_status() == (256, 0)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "e2e_bench_b8_%s_%s.json" % (dtype, inputs)), "w") as f:
        json.dump(errs, f)
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, "B=8 %s %s: %s (all %s)" % (dtype, inputs, bad, errs)


@pytest.mark.parametrize("dtype,force", [("f32", 0), ("f16", 0), ("f16", 5)])
def test_bench_configuration_b64(dtype, force):
    """BASELINE configs[4] regime (north_star "batch 1/8/64"): B=64 streams in one fused step graph.  At this batch the
    engine chooses other kernels than at B<=8 (256x128 tiles, BM=128 halo tiles, merged launches split up,
    `choose_tile` / `halo_choice`); force=5 additionally forces the 256x128 tile for every generic conv.
    Streams are independent, so the oracle runs on 4 sampled streams only (first, last and two inner ones -- last
    exercises the ragged M tail of the big tiles): fp32 vs the float64 Oracle <= 1e-4 with the device-decoded
    best_pscore_id equal to the oracle's per stream, fp16 vs QuantOracle <= 5e-3; Refine is checked at the per-stream
    positions (custom.py:131-154 takes ONE pos for the batch in the reference; per-stream pos = the reference per item)."""
    from oracle.np_oracle import QuantOracle
    from siammask_amd import _lib
    B, sample = 64, [0, 21, 42, 63]
    z = synth.smooth_image_batch(B, 127, stream0=300)
    x = synth.smooth_image_batch(B, 255, stream0=300)
    sd = synth.state_dict("sharp", "synthetic_damped")
    o = Oracle(sd, "sharp") if dtype == "f32" else QuantOracle(sd, "sharp")
    o.template(z[sample].astype(np.float64))
    ocls, oloc, omask = o.track_mask(x[sample].astype(np.float64))
    rng = np.random.default_rng(5)
    twh = rng.uniform(40.0, 110.0, size=(B, 2))
    if force:
        _lib.tune(force_tile=force)
    try:
        m = _model("sharp", "synthetic_damped", dtype, True, max_batch=B)
        m.template(torch.from_numpy(z).cuda())
        out = m.track_step(torch.from_numpy(x).cuda(), torch.from_numpy(twh).cuda(), refine=True)
        torch.cuda.synchronize()
    finally:
        if force:
            _lib.tune(force_tile=0)
    tol = 1e-4 if dtype == "f32" else 5e-3
    errs = {"cls": rel_err(out["cls"][sample].cpu().numpy(), ocls), "loc": rel_err(out["loc"][sample].cpu().numpy(), oloc),
            "mask": rel_err(out["mask"][sample].cpu().numpy(), omask)}
    box = out["box"].cpu().numpy()
    pos = []
    for i, b in enumerate(sample):
        bid, dy, dx, _ = decode_best(ocls[i], oloc[i], target_sz=twh[b], scale_x=1.0)
        if dtype == "f32":
            assert int(box[b, 7]) == bid, "stream %d: device argmax %d != oracle %d" % (b, int(box[b, 7]), bid)
            pos.append((dy, dx))
        else:   # refine at the positions the DEVICE decoded (its fp16 argmax may legitimately differ)
            best = int(box[b, 7])
            pos.append(((best % 625) // 25, best % 25))
    oref = o.track_refine(np.asarray(pos))
    errs["refine"] = rel_err(out["refine"][sample].cpu().numpy(), oref)
    # every stream got its own result (no stream aliasing inside the big tiles): distinct inputs -> distinct logits
    r = out["refine"].cpu().numpy()
    assert len({r[b].tobytes() for b in range(B)}) == B
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "e2e_bench_b64_%s_force%d.json" % (dtype, force)), "w") as f:
        json.dump(errs, f)
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, "B=64 %s force_tile=%d: %s (all %s)" % (dtype, force, bad, errs)


def test_graph_cache_lru_keeps_stable_buffers_hot():
    """serving path (stage=False: graphs keyed on the caller's buffers): a caller that keeps handing over NEW buffers
    must not evict the graph of a caller with a stable buffer (least-recently-used eviction, engine + Python fast path),
    and every call stays correct across evictions"""
    m = _model("sharp", "synthetic_damped", "f16", True)
    z = torch.from_numpy(synth.smooth_image_batch(1, 127, stream0=9)).cuda()
    m.template(z)
    twh = torch.tensor([[60.0, 80.0]], dtype=torch.float64).cuda()
    stable = torch.from_numpy(synth.smooth_image_batch(1, 255, stream0=9)).cuda()
    want = {k: v.clone() for k, v in m.track_step(stable, twh, stage=False).items() if v is not None}
    fresh = [torch.from_numpy(synth.smooth_image_batch(1, 255, stream0=100 + i)).cuda() for i in range(70)]
    for i, x in enumerate(fresh):                     # 70 > 64 graphs, 70 > 32 fast-path entries
        m.track_step(x, twh, stage=False)
        if i % 7 == 0:
            got = m.track_step(stable, twh, stage=False)
            assert all(torch.equal(got[k], want[k]) for k in want), i
    got = m.track_step(stable, twh, stage=False)
    torch.cuda.synchronize()
    assert all(torch.equal(got[k], want[k]) for k in want)
    assert m.seq_status()[1] == 0


def test_persistent_sequences_and_tail_fusion_are_what_runs_at_b8():
    """Guard against a silent fall-back: on an MI355X the placement check of smk_create passes (256 workgroups per sequence
    launch), the B = 8 fp16 frame step really contains 1 conv_seq launch + 1 chain_mask launch in place of 33 + 2 per-layer ones, the device error
    flag stays 0, and switching both features off changes the outputs only by fp16 summation-order noise."""
    from siammask_amd import _lib
    B = 8
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=70)).cuda()
    x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=70)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()

    def run(**knobs):
        _lib.tune(**knobs)
        m = _model("sharp", "synthetic_damped", "f16", True, max_batch=B)
        m.template(z)
        out = {k: v.clone() for k, v in m.track_step(x, twh, refine=True).items() if v is not None}
        m.profile(True)
        m.track_step(x, twh, refine=True)
        recs = m.profile_dump()
        m.profile(False)
        torch.cuda.synchronize()
        return out, recs, m.seq_status()

    try:
        on, recs, (grid, err) = run(seq=1, chain_mask=1)
        # ... and inside the sequence the nine (conv3, next 1x1) pairs -- three in layer2, five in layer3, layer3.5's conv3 with
        # adjust -- run as ONE tile routine each (c3c1_tile.inc): 24 layer records' worth of barriers instead of 33
        fused = _lib.tune_get("seq_fused_last")
        off, recs_off, _ = run(seq=0, chain_mask=0)
        unfused, _, (_, err_u) = run(seq=1, chain_mask=1, seq_fuse=0)
        fused_off = _lib.tune_get("seq_fused_last")
    finally:
        _lib.tune(seq=1, chain_mask=1, seq_fuse=1)
    assert grid == 256 and err == 0 and err_u == 0
    assert fused == 9 and fused_off == 0, (fused, fused_off)
    for k in ("cls", "loc", "mask", "refine"):
        assert rel_err(on[k].cpu().numpy(), unfused[k].cpu().numpy()) <= 5e-3, k
    kernels = [r["kernel"].split("<")[0] for r in recs]
    assert kernels.count("conv_seq") == 1 and kernels.count("chain_mask") == 1, kernels
    # the profiler launches the members of merged launches one by one (per-layer attribution), so its count (30) is
    # above the 24 nodes of the captured graph; what must hold is that 33 convolutions + 2 tail kernels collapsed into 2
    n_on, n_off = sum(r["calls"] for r in recs), sum(r["calls"] for r in recs_off)
    assert n_on + 25 <= n_off and n_on <= 34, (n_on, n_off)
    for k in ("cls", "loc", "mask", "refine"):
        assert rel_err(on[k].cpu().numpy(), off[k].cpu().numpy()) <= 5e-3, k


@pytest.mark.parametrize("B,expect_seq", [(4, False), (5, True), (7, True), (10, False), (12, True), (16, True), (24, True), (32, False)])
def test_which_batches_run_the_persistent_sequence(B, expect_seq):
    """engine.cpp seq_wanted (profiles/r03_seq_batch_sweep.txt, r03h_seq_batch_sweep.txt): the sequence runs where it was measured faster -- B = 5..8, 12, 16, 24 (12: uneven teams, four XCDs with two images) --
    and nowhere else; where it runs, several images per XCD in turn (B = 16, 24) or idle XCDs (B = 6, 7) give the per-launch path's outputs
    up to fp16 summation order, and the device error flag stays 0."""
    from siammask_amd import _lib
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=11)).cuda()
    x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=11)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()

    def run(**knobs):
        _lib.tune(**knobs)
        m = _model("sharp", "synthetic_damped", "f16", True, max_batch=B)
        m.template(z)
        out = {k: v.clone() for k, v in m.track_step(x, twh, refine=True).items() if v is not None}
        m.profile(True)
        m.track_step(x, twh, refine=True)
        recs = m.profile_dump()
        m.profile(False)
        torch.cuda.synchronize()
        return out, [r["kernel"].split("<")[0] for r in recs], m.seq_status()

    try:
        on, kernels, (grid, err) = run(seq=1)
        assert err == 0
        assert kernels.count("conv_seq") == (1 if expect_seq else 0), (B, kernels)
        if expect_seq:
            off, kernels_off, _ = run(seq=0)
            assert kernels_off.count("conv_seq") == 0
            for k in ("cls", "loc", "mask", "refine"):
                assert rel_err(on[k].cpu().numpy(), off[k].cpu().numpy()) <= 5e-3, (B, k)
    finally:
        _lib.tune(seq=1)


@pytest.mark.parametrize("B", [9, 10, 13])
def test_uneven_teams_do_not_write_over_each_other(B):
    """the persistent sequence forced on for batches where some teams own two images and some one (the teams are not synchronised
    with each other; the one-image teams run a stage ahead): every intermediate of layer2 / layer3 has its own buffer with ONE
    layout, so p2 / p3 / search agree with the per-launch path for every image.  (With layer2.0's 63x63 conv1 output and the 31x31
    conv1 outputs in one buffer, images 0 / 1 of the slow teams were overwritten: profiles/r03h_b12_race.txt.)"""
    from siammask_amd import _lib
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=11)).cuda()
    x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=11)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()

    def run(**knobs):
        _lib.tune(**knobs)
        m = _model("sharp", "synthetic_damped", "f16", True, max_batch=B)
        m.template(z)
        m.track_step(x, twh, refine=True)
        out = {n: m.debug_tensor(n).cpu().numpy().astype(np.float64) for n in ("p2", "p3", "search")}
        st = m.seq_status()
        return out, st

    try:
        off, _ = run(seq=0)
        on, (grid, err) = run(seq=1, seq_min_batch=1, seq_max_batch=64)
    finally:
        _lib.tune(seq=1, seq_min_batch=5, seq_max_batch=8)
    assert grid == 256 and err == 0
    for n in on:
        per = [rel_err(on[n][b], off[n][b]) for b in range(B)]
        assert max(per) <= 5e-3, (B, n, ["%.0e" % v for v in per])


@pytest.mark.parametrize("B", [8, 16])
def test_fused_step_is_deterministic_over_replays(B):
    """a race between workgroups of the persistent sequence (team barriers, early residual fetch of the fused pairs, patch double
    buffer) would show as run-to-run differences: 40 replays of the fused frame step give the same bits, at one image per team
    (B = 8: pairs fused, patch-sharing tiles) and at two (B = 16: patch-sharing tiles)"""
    m = _model("sharp", "synthetic_damped", "f16", True, max_batch=B)
    z = torch.from_numpy(synth.image_batch(B, 127, stream0=7)).cuda()
    x = torch.from_numpy(synth.image_batch(B, 255, stream0=7)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
    m.template(z)
    first = {k: v.clone() for k, v in m.track_step(x, twh, refine=True).items() if v is not None}
    p2 = m.debug_tensor("p2").clone()
    for it in range(40):
        out = m.track_step(x, twh, refine=True)
        for k in first:
            assert torch.equal(out[k], first[k]), (B, it, k)
    assert torch.equal(m.debug_tensor("p2"), p2)
    assert m.seq_status() == (256, 0)


def test_producer_variants_bit_equal_end_to_end():
    """smk_tune a_stage (activation rows of conv_wreg_kernel through registers instead of LDS-DMA; since the end of round 4 the persistent
    sequence kernel no longer carries that arm -- every routine it carries costs the others registers -- and ignores the knob) and npw
    (two or four producer waves) change the data path of the producers only: the fused B = 8 frame step (persistent sequences on) must
    give bit-identical outputs for every combination, and the device error flag of the sequences stays 0."""
    from siammask_amd import _lib
    B = 8
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=90)).cuda()
    x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=90)).cuda()
    twh = torch.tensor([[70.0, 50.0]] * B, dtype=torch.float64).cuda()
    outs = []
    saved = {k: _lib.tune_get(k) for k in ("a_stage", "npw")}
    variants = ((0, 2), (1, 2), (0, 4), (1, 4))
    try:
        for a, n in variants:
            _lib.tune(a_stage=a, npw=n)
            m = _model("sharp", "synthetic_damped", "f16", True, max_batch=B)
            m.template(z)
            m.track_step(x, twh, refine=True)
            o = m.track_step(x, twh, refine=True)                  # second call = graph replay
            torch.cuda.synchronize()
            outs.append({k: v.clone() for k, v in o.items() if v is not None})
            assert m.seq_status() == (256, 0), (a, n, m.seq_status())
            del m
    finally:
        _lib.tune(**saved)
    for v, o in zip(variants[1:], outs[1:]):
        for k in outs[0]:
            assert torch.equal(outs[0][k], o[k]), (v, k)
