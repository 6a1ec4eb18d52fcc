"""Pin the numpy oracle (oracle/np_oracle.py) against outputs of the reference itself.

The reference has no golden vectors of its own; the fixtures under
tests/golden/ were produced by oracle/make_golden.py running the reference modules in
float64 on CPU.  Tolerance: the oracle runs in float64 too, so agreement is at float64
round-off amplified by the network (<= 1e-9 rel-to-max; fixtures are stored as float32,
hence the 2e-7 storage floor)."""
import numpy as np
import pytest

from oracle.np_oracle import Oracle, decode_best
from siammask_amd import synth
from helpers import CASES, load_golden, rel_err, sampled_err

TOL = 5e-7  # float32 storage of the fixtures dominates


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference(case):
    g = load_golden(case)
    variant, fixture = str(g["variant"]), str(g["fixture"])
    o = Oracle(synth.state_dict(variant, fixture), variant)
    o.template(g["z_u8"].astype(np.float64))
    assert rel_err(o.zf, g["zf_full"]) < TOL
    x = g["x_u8"].astype(np.float64)
    if variant == "rpn":
        cls, loc = o.track(x)
    else:
        cls, loc, mask = o.track_mask(x)
        assert sampled_err(g, "mask", mask) < TOL
        for b in range(x.shape[0]):
            y_, x_ = g["best_yx"][b]
            assert rel_err(mask[b, :, y_, x_], g["mask_col"][b]) < TOL
        assert sampled_err(g, "search", o.search) < TOL
        assert sampled_err(g, "corr_mask", o.corr_feature) < TOL
    assert rel_err(cls, g["cls"]) < TOL
    assert rel_err(loc, g["loc"]) < TOL
    for b in range(x.shape[0]):
        bid, dy, dx, _ = decode_best(cls[b], loc[b])
        assert bid == int(g["best_id"][b])           # bit-exact argmax box index
        assert (dy, dx) == tuple(int(v) for v in g["best_yx"][b])
    if variant == "sharp":
        for n_, f_ in zip(("p0", "p1", "p2", "p3"), o.feature):
            assert sampled_err(g, n_, f_) < TOL
        ref = o.track_refine(g["best_yx"])
        assert rel_err(ref, g["refine"]) < TOL
        shared = o.track_refine(tuple(int(v) for v in g["shared_pos"]))
        assert rel_err(shared, g["refine_shared"]) < TOL


def test_fixture_is_well_conditioned():
    """Logits O(1) and an argmax gap far above the fp32 tolerance."""
    for case in CASES:
        g = load_golden(case)
        assert np.abs(g["cls"]).max() < 50 and np.abs(g["loc"]).max() < 50
        assert g["top2_gap"].min() > 1e-3


def test_torch_port_matches_reference():
    """The CPU-baseline port (oracle/torch_port.py, fp32) reproduces the reference's outputs
    within the fp32 noise floor of the fixture (<= 8.4e-6 vs float64)."""
    import torch
    from oracle.torch_port import TorchPort
    g = load_golden("sharp_damped_b2")
    t = TorchPort(synth.state_dict("sharp", "synthetic_damped"), "sharp")
    with torch.no_grad():
        t.template(torch.from_numpy(g["z_u8"].astype(np.float32)))
        cls, loc, mask = t.track_mask(torch.from_numpy(g["x_u8"].astype(np.float32)))
        assert rel_err(cls.numpy(), g["cls"]) < 1e-4
        assert rel_err(loc.numpy(), g["loc"]) < 1e-4
        assert sampled_err(g, "mask", mask.numpy()) < 1e-4
        shared = t.track_refine(tuple(int(v) for v in g["shared_pos"]))
        assert rel_err(shared.numpy(), g["refine_shared"]) < 1e-4


def test_quant_oracle_tracks_the_reference():
    """The quantisation-aware oracle (fp16 rounding points of the device path) stays within the loose fp16
    gates of the reference outputs, and its two Refine summation orders (which branch is stored before the
    sum: the chain kernel stores v*.2, the per-layer path h*.2) differ by fp16 round-off only."""
    from oracle.np_oracle import QuantOracle
    g = load_golden("sharp_damped_b2")        # the fixture the fp16 GPU gates use
    sd = synth.state_dict("sharp", str(g["fixture"]))
    z, x = g["z_u8"].astype(np.float64), g["x_u8"].astype(np.float64)
    refs = {}
    for order in (True, False):
        q = QuantOracle(sd, "sharp", refine_sum_in_h=order)
        q.template(z)
        cls, loc, mask = q.track_mask(x)
        refs[order] = q.track_refine(g["best_yx"])
    assert rel_err(cls, g["cls"]) < 3e-2 and rel_err(loc, g["loc"]) < 3e-2
    assert rel_err(refs[True], g["refine"]) < 1e-2 and rel_err(refs[False], g["refine"]) < 1e-2
    assert 0 < rel_err(refs[True], refs[False]) < 5e-3


def test_summation_order_alone_moves_the_fp16_outputs_by_the_tight_gate():
    """Why the fp16 gate against the quantisation-aware oracle is 5e-3 and not the 2e-3 SURVEY.md proposed (round-1
    verdict, item 10).  The SAME quantised network (fp16 weights, fp16 stored activations) evaluated with three summation
    models -- exact sums, a float32 accumulator fed one 16-element k-step at a time in the device's K order
    (conv2d_f32acc), the same with two interleaved accumulators (a K split inside the workgroup) -- differs from itself by
    ~5e-4 after the stem, ~1.5e-3 after layer2 and ~3e-3 at `search`: every flipped fp16 rounding is amplified by the
    network.  The kernels use several orders (tile shape and K split per layer and batch size), so no oracle order can pin
    the device below that floor; the measured device-vs-oracle errors (3e-4 ... 3.4e-3) sit right on it.
    The argmax of the decode is the same for all models."""
    from oracle.np_oracle import QuantOracle, decode_best
    sd = synth.state_dict("sharp", "synthetic_damped")
    z = synth.smooth_image_batch(1, 127, stream0=5).astype(np.float64)
    x = synth.smooth_image_batch(1, 255, stream0=5).astype(np.float64)
    runs = {}
    for name, kw in (("exact", {}), ("f32", {"accum": "f32"}), ("f32x2", {"accum": "f32", "ksplit": 2})):
        q = QuantOracle(sd, "sharp", **kw)
        q.template(z)
        cls, loc, mask = q.track_mask(x)
        runs[name] = dict(cls=cls, loc=loc, mask=mask, refine=q.track_refine((12, 12)), search=q.search,
                          p0=q.feature[0], p2=q.feature[2], bid=decode_best(cls[0], loc[0])[0])
    assert len({r["bid"] for r in runs.values()}) == 1
    for a, b in (("exact", "f32"), ("f32", "f32x2")):
        e = {k: rel_err(runs[b][k], runs[a][k]) for k in ("p0", "p2", "search", "cls", "loc", "mask", "refine")}
        assert 1e-4 <= e["p0"] <= 1e-3, (a, b, e)                   # one layer: a few flipped roundings
        assert 1.5e-3 <= e["search"] <= 5e-3, (a, b, e)             # ... amplified through 50 layers: above 2e-3, below the gate
        assert all(v <= 5e-3 for v in e.values()), (a, b, e)
