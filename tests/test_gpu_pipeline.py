"""Software-pipelined frame steps (smk_set_pipeline, ABI 1.5).

The reference's tracker crops frame f + 1 at the box decoded from frame f (/root/reference/tools/test.py:240-250,302-308);
the Refine mask (:257-284) is an output nothing on the device waits for.  The pipelined step runs Refine (+ the 63x63 mask
head) of frame f on a side stream beside stem + layer1 of frame f + 1.  It must change WHEN things run and nothing else:
every output and every ring row bit-identical to the serial step."""
import ctypes

import numpy as np
import pytest
import torch

from siammask_amd import _lib, spec, synth

pytestmark = pytest.mark.gpu


def _model(B, dtype="f16"):
    from siammask_amd.custom import build
    m = build("sharp", dtype=dtype, graph=True, max_batch=B)
    m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
    return m.eval().cuda()


def _inputs(B, n, seed):
    z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=seed)).cuda()
    xs = [torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=seed + 7 * i)).cuda() for i in range(n)]
    g = np.random.Generator(np.random.PCG64(seed))
    twh = torch.from_numpy(g.uniform(40.0, 110.0, size=(B, 2))).cuda()
    return z, xs, twh


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("B,dtype,frames", [(8, "f16", 7), (64, "f16", 4), (1, "f16", 5), (8, "f32", 3), (3, "f16", 4), (16, "f16", 4), (24, "f16", 3),
                                            (8, "f16x3", 3), (1, "f16x3", 3)])
def test_pipelined_rows_equal_serial_rows(B, dtype, frames, depth):
    """ring rows (box f64 + fp16 Refine logits) and the step's own outputs, frame by frame, serial vs pipelined -- B = 8 / 16 / 24 run
    layer2 .. adjust as the persistent sequence (which waits for the tail; depth 2 cuts the tail in two there, B = 24 without the
    mask head in the chain launch), B = 64 / 1 / 3 the per-launch kernels (main gate in front of the heads), fp32 ends its Refine in
    the stand-alone ring commit launch; the split-operand contexts (f16x3) run per-launch kernels with the fp16 chain tail"""
    m = _model(B, dtype)
    z, xs, twh = _inputs(B, frames, 500 + B)
    m.template(z)
    box, ref = m.set_result_ring(frames, batch=B)
    outs = []
    for x in xs:
        o = m.track_step(x, twh, refine=True, stage=False)
        torch.cuda.synchronize()
        outs.append({k: o[k].clone() for k in ("box", "refine", "cls", "loc", "mask")})
    want_box, want_ref = box.clone(), ref.clone()
    assert m.result_ring_frames(reset=True) == frames
    box.zero_(); ref.zero_()

    m.set_pipeline(depth)
    got = []
    for i, x in enumerate(xs):
        o = m.track_step(x, twh, refine=True, stage=False)
        # box / cls / loc of THIS frame are complete in stream order; refine / mask behind the join (depth 2: which launches the
        # second part of the tail without its gate)
        cur = {k: o[k].clone() for k in ("box", "cls", "loc")}
        m.pipeline_join()
        cur.update({k: o[k].clone() for k in ("refine", "mask")})
        got.append(cur)
    assert m.result_ring_frames() == frames
    torch.cuda.synchronize()
    for i in range(frames):
        for k in ("box", "cls", "loc", "refine", "mask"):
            assert torch.equal(got[i][k], outs[i][k]), (i, k)
    assert torch.equal(box, want_box)
    assert torch.equal(ref, want_ref)
    g, e = m.seq_status()
    assert e == 0

    # free-running (no join between steps): the rows must still be the serial rows
    assert m.result_ring_frames(reset=True) == frames
    box.zero_(); ref.zero_()
    for x in xs:
        m.track_step(x, twh, refine=True, stage=False)
    assert m.result_ring_frames() == frames
    assert torch.equal(box, want_box)
    assert torch.equal(ref, want_ref)
    # ... and the step's own output buffers hold the LAST frame's results behind the join
    m.pipeline_join()
    torch.cuda.synchronize()
    for k in ("refine", "mask"):
        assert torch.equal(o[k], outs[-1][k]), k
    # mixed forms on one ring: serial, pipelined, without Refine -- both ring cursors advance once per frame whatever the form
    assert m.result_ring_frames(reset=True) == frames
    m.track_step(xs[0], twh, refine=True, stage=False)
    m.set_pipeline(0)
    m.track_step(xs[1], twh, refine=True, stage=False)
    m.set_pipeline(depth)
    m.track_step(xs[2], twh, refine=True, stage=False)
    assert m.result_ring_frames() == 3
    assert torch.equal(box[:3], want_box[:3]) and torch.equal(ref[:3], want_ref[:3])

    # serial entry points behind a pipelined step: they join the tail and see that frame's features
    o = m.track_step(xs[0], twh, refine=True, stage=False)
    pos = torch.tensor([[12, 12]] * B, dtype=torch.int32).cuda()
    r1 = m.track_refine(pos).clone()
    m.set_pipeline(0)
    m.track_step(xs[0], twh, refine=True, stage=False)
    r0 = m.track_refine(pos).clone()
    torch.cuda.synchronize()
    assert torch.equal(r0, r1)


@pytest.mark.parametrize("knobs", [dict(pipe_join=0), dict(pipe_sig=0), dict(pipe_sig=1), dict(pipe_eager=2), dict(pipe_two_form=0), dict(pipe_two_form=2)])
def test_other_join_forms_are_the_same_arithmetic(knobs):
    """the measured alternatives of the two joins, kept for the A/B (profiles/r05a_*, r05e_*, r05f_*): pipe_join = 0 -- three graphs and
    a cross-queue event wait instead of the in-stream gate kernel; pipe_sig = 0 -- the tail's start by event record + wait instead
    of the gate that polls beside the persistent launch; pipe_sig = 1 -- hipStreamWaitValue32 on signal memory; pipe_eager = 2 -- the
    tail as eager launches; pipe_two_form = 0 / 2 -- depth 2 with chain + mask head as ONE launch behind the persistent launch /
    behind conv_search (profiles/r05j_*, r05o_*).  Same rows."""
    if not _lib.tune_get("measure_build"):
        # round 6: the forms that measured slower are compiled by `make MEASURE=1` only; the product library refuses them
        with pytest.raises(RuntimeError):
            _lib.tune(**knobs)
        pytest.skip("the measured alternatives of the pipelined step are only in a library built with `make MEASURE=1`")
    old = {k: _lib.tune_get(k) for k in knobs}
    try:
        _lib.tune(**knobs)
        test_pipelined_rows_equal_serial_rows(8, "f16", 5, 2 if ("pipe_eager" in knobs or "pipe_two_form" in knobs) else 1)
    finally:
        _lib.tune(**old)


@pytest.mark.parametrize("depth", [1, 2])
def test_pipelined_200_steps_clean_and_deterministic(depth):
    """200 free-running pipelined steps at the bench configuration: the persistent sequence never reports a failure (it waits
    for the tail, so it still owns every CU -- with the tail's gate wave resident beside it), the frame counter arrives at 200, and
    the last rows equal a second run's"""
    B, rows = 8, 4
    m = _model(B)
    z, xs, twh = _inputs(B, 4, 900)
    m.template(z)
    box, ref = m.set_result_ring(rows, batch=B)
    m.set_pipeline(depth)
    snaps = []
    for rep in range(2):
        for i in range(200):
            m.track_step(xs[i % 4], twh, refine=True, stage=False)
        assert m.result_ring_frames(reset=True) == 200
        snaps.append((box.clone(), ref.clone()))
        g, e = m.seq_status()
        assert g > 0 and e == 0, (g, e)
    assert torch.equal(snaps[0][0], snaps[1][0]) and torch.equal(snaps[0][1], snaps[1][1])
    assert m.seq_recovered == 0


def test_a_slow_stream_in_front_of_the_step_is_not_a_gate_timeout():
    """the gate at the head of a tail starts polling as soon as the side stream is free -- possibly long before its step starts on the
    caller's stream.  Half a second of somebody else's work in front of the step must not trip it (its limit is 5 s; the main gate's 0.2 s
    bounds a wait for a tail that is already running)"""
    B = 8
    m = _model(B)
    z, xs, twh = _inputs(B, 2, 970)
    m.template(z)
    want = []
    for x in xs:
        o = m.track_step(x, twh, refine=True, stage=False)
        torch.cuda.synchronize()
        want.append({k: v.clone() for k, v in o.items() if v is not None})
    m.set_pipeline(1)
    for i, x in enumerate(xs):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        torch.cuda._sleep(int(1.0e9))                       # ~0.4-0.6 s of device time on the step's own stream
        t1.record()
        o = m.track_step(x, twh, refine=True, stage=False)
        m.pipeline_join()
        torch.cuda.synchronize()
        for k in want[i]:
            assert torch.equal(o[k], want[i][k]), (i, k)
        if t0.elapsed_time(t1) < 250.0:                     # (shorter than the 0.2 s limit it is meant to exceed: correct, but it proved nothing)
            pytest.skip("torch.cuda._sleep took only %.0f ms on this box" % t0.elapsed_time(t1))
    g, e = m.seq_status()
    assert g > 0 and e == 0 and m.seq_recovered == 0


def test_a_stream_blocked_for_longer_than_the_gate_limit_only_delays_the_step():
    """the tail gate's 5 s used to count from the moment the side stream was free.  A caller's stream that is blocked in front of
    the step for longer than that -- a collective waiting for a straggler rank, a host-fed event, a large copy; here ~6 s of spinning -- made
    the gate give the frame up (code 3: SMK_E_SEQ, template invalidated, serial steps for the rest of the run).  Now the clock is armed by the
    step's own main gate ("this step's main part is running", pipe_cnt[8]): the frame is late, right, and the context stays pipelined."""
    B = 8
    m = _model(B)
    z, xs, twh = _inputs(B, 3, 990)
    m.template(z)
    want = []
    for x in xs:
        o = m.track_step(x, twh, refine=True, stage=False)
        torch.cuda.synchronize()
        want.append({k: v.clone() for k, v in o.items() if v is not None})
    m.set_pipeline(1)
    outs = [m.track_step(xs[0], twh, refine=True, stage=False)]           # a normal pipelined frame first: the word is lowered again behind it
    m.pipeline_join()
    outs[0] = {k: v.clone() for k, v in outs[0].items() if v is not None}
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(14):
        torch.cuda._sleep(int(1.0e9))                                     # ~6 s of device time on the step's own stream, in front of the step
    t1.record()
    for x in xs[1:]:
        o = m.track_step(x, twh, refine=True, stage=False)                # free-running behind the blocked stream
        outs.append(o)
    m.pipeline_join()
    torch.cuda.synchronize()
    for k in want[2]:
        assert torch.equal(outs[2][k], want[2][k]), k
    for k in ("cls", "loc", "box"):                                       # frame 1's box outputs (its mask / refine buffers were re-used by frame 2)
        assert torch.equal(outs[0][k], want[0][k]), k
    g, e = m.seq_status()
    assert g > 0 and e == 0 and m.seq_recovered == 0
    if t0.elapsed_time(t1) < 5200.0:
        pytest.skip("the spin took only %.0f ms on this box: right, but shorter than the 5 s it is meant to exceed" % t0.elapsed_time(t1))


def test_gate_timeout_is_loud_and_leaves_serial_steps_behind():
    """a gate that waits 0.2 s for its partner (a profiler that serialises the two queues produces exactly this) raises failure code 3: the
    next entry point returns SMK_E_SEQ, the context goes back to serial steps, and after template() the frames are right again -- here the
    flag is raised by hand"""
    B = 8
    m = _model(B)
    z, xs, twh = _inputs(B, 2, 950)
    m.template(z)
    want = {k: v.clone() for k, v in m.track_step(xs[0], twh, refine=True, stage=False).items() if v is not None}
    torch.cuda.synchronize()
    m.set_pipeline(1)
    m.track_step(xs[1], twh, refine=True, stage=False)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().smk_debug_seq_inject(m._ctx, 3))
    with pytest.raises(_lib.SmkError) as ei:
        m.track_step(xs[0], twh, refine=True, stage=False)
    assert ei.value.code == _lib.E_SEQ and "gate" in str(ei.value)
    m.template(z)
    o = m.track_step(xs[0], twh, refine=True, stage=False)       # serial now (the library switched pipelining off)
    torch.cuda.synchronize()                                     # (no join needed)
    for k in want:
        assert torch.equal(o[k], want[k]), k
    g, e = m.seq_status()
    assert g > 0                                                 # the persistent sequence itself is still in use


def test_ring_batch_is_recorded_by_the_library():
    """a C caller that sets a ring for one batch and steps another one must get SMK_E_ARG, not rows
    written past the ring"""
    m = _model(8)
    z = torch.from_numpy(synth.smooth_image_batch(4, 127, stream0=310)).cuda()
    m.template(z)
    box = torch.zeros((2, 2, 8), dtype=torch.float64).cuda()           # sized for batch 2
    L = _lib.lib()
    _lib.check(L.smk_set_result_ring(m._ctx, box.data_ptr(), None, 2, 2))
    x = torch.from_numpy(synth.smooth_image_batch(4, 255, stream0=310)).cuda()
    twh = torch.tensor([[60.0, 80.0]] * 4, dtype=torch.float64).cuda()
    cls = torch.empty((4, 10, 25, 25)).cuda(); loc = torch.empty((4, 20, 25, 25)).cuda()
    b8 = torch.empty((4, 8), dtype=torch.float64).cuda()
    rc = L.smk_step(m._ctx, x.data_ptr(), 4, _lib.TRACK_MASK | _lib.TRACK_NO_MASK_HEAD, twh.data_ptr(), cls.data_ptr(),
                    loc.data_ptr(), None, b8.data_ptr(), None, _lib.current_stream_ptr())
    assert rc == -1 and b"result ring was set for batch 2" in L.smk_last_error()
    assert L.smk_set_result_ring(m._ctx, box.data_ptr(), None, 2, 9) == -1     # beyond max_batch
    _lib.check(L.smk_set_result_ring(m._ctx, None, None, 0, 0))
    torch.cuda.synchronize()


def test_pipelined_ring_rows_against_the_oracle():
    """VERDICT r5 (weak #3): every other test of this file compares pipelined rows with serial rows, and the serial step with the oracle
    elsewhere.  Here the ring rows of three free-running PIPELINED frames at the bench's batch are held against the oracle directly:
    fp16 context vs the quantisation-aware oracle (the fp16 gate of tests/test_gpu_e2e.py, 5e-3), the Refine logits at the positions
    the device decoded (its fp16 argmax may legitimately differ from the fp64 one), the box row's index / score against
    decode_best of the oracle's own cls / loc on the streams where both pick the same anchor."""
    from oracle.np_oracle import QuantOracle, decode_best
    from helpers import rel_err
    B, frames = 8, 3
    m = _model(B, "f16")
    z, xs, twh = _inputs(B, frames, 731)
    m.template(z)
    box, ref = m.set_result_ring(frames, batch=B)
    m.set_pipeline(1)
    for x in xs:
        m.track_step(x, twh, refine=True, stage=False)       # free-running: no synchronisation between the frames
    m.pipeline_join()
    torch.cuda.synchronize()
    assert m.result_ring_frames() == frames and m.seq_status() == (256, 0)
    o = QuantOracle(synth.state_dict("sharp", "synthetic_damped"), "sharp")
    o.template(z.cpu().numpy().astype(np.float64))
    twh_h = twh.cpu().numpy()
    same = 0
    for f, x in enumerate(xs):
        ocls, oloc, _ = o.track_mask(x.cpu().numpy().astype(np.float64))
        row = box[f].cpu().numpy()
        best = row[:, 7].astype(np.int64)
        oref = o.track_refine(np.stack([(best % 625) // 25, best % 25], 1))
        e = rel_err(ref[f].float().cpu().numpy().reshape(B, -1), oref.reshape(B, -1))
        assert e <= 5e-3, "frame %d: pipelined ring logits vs the oracle %.2e" % (f, e)
        for b in range(B):
            bid, _, _, ps = decode_best(ocls[b], oloc[b], target_sz=twh_h[b], scale_x=1.0)
            if bid == best[b]:
                same += 1
                assert abs(row[b, 6] - ps[bid]) <= 5e-3 * max(1.0, abs(ps[bid])), (f, b, row[b, 6], ps[bid])
            else:       # an fp16 pick: still one of the oracle's near-best candidates
                assert ps[best[b]] >= ps[bid] - 2e-2, (f, b, best[b], bid, ps[best[b]], ps[bid])
    assert same >= int(0.8 * B * frames), same
