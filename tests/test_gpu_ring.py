"""Result ring (smk_set_result_ring; SURVEY.md 8e "gather of boxes / masks only at the end of a batch of frames",
/root/reference/tools/test.py:296-311 keeps box + mask per frame): with a ring set, the frame step itself writes its decoded
box and its fp16 Refine logits into row (frame % rows) -- the rows bench.py gathers -- and nothing else about the step changes."""
import numpy as np
import pytest
import torch

from siammask_amd import spec, synth

pytestmark = pytest.mark.gpu


def _model(B, dtype="f16"):
    from siammask_amd.custom import build
    m = build("sharp", dtype=dtype, graph=True, max_batch=B)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    return m.eval().cuda()


@pytest.mark.parametrize("B,rows", [(8, 3), (1, 4), (5, 2)])
def test_ring_rows_are_the_frames_results(B, rows):
    m = _model(B)
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=300)).cuda()
    xs = [torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=300 + 7 * i)).cuda() for i in range(5)]
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
    m.template(z)
    want = []
    for x in xs:                                      # no ring: the plain step
        o = m.track_step(x, twh, refine=True, stage=False)
        want.append({k: o[k].clone() for k in ("box", "refine", "cls", "loc", "mask")})
    torch.cuda.synchronize()
    box, ref = m.set_result_ring(rows, batch=B)
    assert tuple(box.shape) == (rows, B, 8) and tuple(ref.shape) == (rows, B, spec.REFINE_OUT ** 2)
    for i, x in enumerate(xs):
        o = m.track_step(x, twh, refine=True, stage=False)
        for k in ("box", "refine", "cls", "loc", "mask"):        # the step's own outputs are untouched by the commit launch
            assert torch.equal(o[k], want[i][k]), (i, k)
        torch.cuda.synchronize()
        r = i % rows
        assert torch.equal(box[r], want[i]["box"]), i
        assert torch.equal(ref[r], want[i]["refine"].half()), i
        assert m.result_ring_frames() == i + 1
    # rows that were not overwritten still hold their frames (5 frames into `rows` rows: frame f lives in row f % rows)
    for f in range(len(xs) - rows, len(xs)):
        assert torch.equal(box[f % rows], want[f]["box"])
    assert m.result_ring_frames(reset=True) == len(xs) and m.result_ring_frames() == 0
    m.track_step(xs[0], twh, refine=True, stage=False)
    torch.cuda.synchronize()
    assert torch.equal(box[0], want[0]["box"]) and m.result_ring_frames() == 1
    # off again: the step no longer writes the rows
    m.set_result_ring(0)
    box.zero_()
    m.track_step(xs[1], twh, refine=True, stage=False)
    torch.cuda.synchronize()
    assert float(box.abs().sum()) == 0.0


@pytest.mark.parametrize("dtype,ring_refine,step_refine", [("f32", True, True), ("f16", False, True), ("f16", True, False)])
def test_ring_other_step_shapes(dtype, ring_refine, step_refine):
    """the ring writes travel with other launches depending on the step: fp32's Refine does not end in the chain kernel (its
    logits take the stand-alone commit launch), a boxes-only ring, and a step without Refine (the decode launch advances the
    cursor itself) -- rows and frame count must come out the same"""
    B, rows = 4, 3
    m = _model(B, dtype)
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=330)).cuda()
    xs = [torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=330 + 7 * i)).cuda() for i in range(4)]
    twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
    m.template(z)
    want = []
    for x in xs:
        o = m.track_step(x, twh, refine=step_refine, stage=False)
        want.append({k: o[k].clone() for k in ("box", "refine") if o[k] is not None})
    torch.cuda.synchronize()
    box, ref = m.set_result_ring(rows, batch=B, refine=ring_refine)
    assert (ref is not None) == ring_refine
    for i, x in enumerate(xs):
        o = m.track_step(x, twh, refine=step_refine, stage=False)
        torch.cuda.synchronize()
        assert torch.equal(o["box"], want[i]["box"])
        assert torch.equal(box[i % rows], want[i]["box"]), i
        if ring_refine and step_refine:
            assert torch.equal(ref[i % rows], want[i]["refine"].half()), i
        assert m.result_ring_frames() == i + 1, (i, m.result_ring_frames())


def test_ring_needs_its_batch():
    m = _model(8)
    z = torch.from_numpy(synth.smooth_image_batch(4, 127, stream0=310)).cuda()
    m.template(z)
    m.set_result_ring(2, batch=8)
    x = torch.from_numpy(synth.smooth_image_batch(4, 255, stream0=310)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * 4, dtype=torch.float64).cuda()
    with pytest.raises(ValueError):
        m.track_step(x, twh)
