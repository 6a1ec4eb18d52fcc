"""The end-of-batch gather beside the next batch of frames, on ONE GPU.

DESIGN.md 7 promises that the RCCL gather of a batch's results runs on a side stream while the next batch of frames -- whose
step contains the persistent conv_seq_kernel launch that wants every CU -- is already running; bench.py itself avoids the
overlap (it gathers once, after the timed loop).  Here a 1-rank `nccl` process group sends the gather through RCCL anyway
(`ResultGather(always_collective=True)`), the next frames are enqueued without waiting for it, and everything must come out
right: the gathered rows are the rows that were written, the frames computed beside the gather are bit-identical to frames
computed alone, and the sequence kernel reports no barrier time-out."""
import socket

import pytest
import torch

from siammask_amd import dist as sdist
from siammask_amd import spec, synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_side_stream_gather_overlaps_the_next_frames_with_the_persistent_sequence():
    import torch.distributed as dist
    from siammask_amd.custom import build
    B, K, ROUNDS = 8, 4, 5
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        m = build("sharp", dtype="f16", max_batch=B, graph=True)
        m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
        m = m.eval().cuda()
        z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=900)).cuda()
        xs = [torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=900 + 10 * i)).cuda() for i in range(K)]
        twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
        m.template(z)
        # the frames alone (no gather anywhere near them)
        want = []
        for x in xs:
            o = m.track_step(x, twh, refine=True, stage=False)
            want.append((o["box"].clone(), o["refine"].clone()))
        torch.cuda.synchronize()
        assert m.seq_status() == (256, 0), "this test is about the persistent sequence launch"
        g = sdist.ResultGather(dev, always_collective=True)
        rows = [(torch.empty((K, B, 8), dtype=torch.float64, device=dev),
                 torch.empty((K, B, spec.REFINE_OUT ** 2), dtype=torch.float16, device=dev)) for _ in range(2)]
        pending = None
        for it in range(ROUNDS):
            box, masks = rows[it % 2]                         # double-buffered result rows: the gather of round it-1 still reads the other pair
            for k, x in enumerate(xs):
                o = m.track_step(x, twh, refine=True, stage=False)
                box[k].copy_(o["box"])
                masks[k].copy_(o["refine"])
            if pending is not None:                          # results of the PREVIOUS round: gathered while this round's frames ran
                g.wait()
                gb, gm = pending
                for k in range(K):
                    assert torch.equal(gb[0, k], want[k][0]), (it, k)
                    assert torch.equal(gm[0, k].float(), want[k][1].half().float()), (it, k)
            pending = g.gather(box, masks)                   # side stream; NOT waited for: the next round starts right away
        g.wait()
        gb, gm = pending
        torch.cuda.synchronize()
        for k in range(K):
            assert torch.equal(gb[0, k], want[k][0]) and torch.equal(gm[0, k].float(), want[k][1].half().float())
        grid, err = m.seq_status()
        assert err == 0 and grid == 256, "the gather beside a sequence launch must never cost it its co-residency (grid %d, err %d)" % (grid, err)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("extra_streams", [0, 8])
def test_pipelined_steps_beside_the_streams_of_a_multi_rank_run(extra_streams):
    """bench.py at --gpus N > 1 has more live streams than the single-GPU tests: torch's null stream, the pipeline's side stream,
    the process group's RCCL stream (barrier / all_reduce), the gather's side stream.  HIP multiplexes streams onto a handful of
    hardware queues; the two gate kernels of a pipelined step wait for EACH OTHER's streams, so the caller's stream and the
    pipeline's side stream must not end up in one queue (that is a gate time-out: SMK_E_SEQ, serial steps from there on -- loud,
    but a wrong figure for the run).  Here: a 1-rank `nccl` group with every one of those streams live (plus, second case, eight
    more busy user streams), then free-running pipelined steps: no time-out, rows equal the serial rows."""
    import torch.distributed as dist
    from siammask_amd.custom import build
    B, K = 8, 4
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        dist.barrier()
        t = torch.ones(4, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g = sdist.ResultGather(dev, always_collective=True)
        g.gather(torch.zeros((2, 8), dtype=torch.float64, device=dev))
        g.wait()
        users = [torch.cuda.Stream(device=dev) for _ in range(extra_streams)]
        junk = torch.zeros(1 << 20, device=dev)
        for s in users:
            with torch.cuda.stream(s):
                junk.add_(1.0)
        torch.cuda.synchronize()
        m = build("sharp", dtype="f16", max_batch=B, graph=True)
        m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
        m = m.eval().cuda()
        z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=930)).cuda()
        xs = [torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=930 + 10 * i)).cuda() for i in range(K)]
        twh = torch.tensor([[60.0, 80.0]] * B, dtype=torch.float64).cuda()
        m.template(z)
        box, ref = m.set_result_ring(K, batch=B)
        for x in xs:
            m.track_step(x, twh, refine=True, stage=False)
        assert m.result_ring_frames(reset=True) == K
        want = (box.clone(), ref.clone())
        box.zero_(); ref.zero_()
        m.set_pipeline(1)
        for rep in range(10):
            for x in xs:
                m.track_step(x, twh, refine=True, stage=False)
                for s in users[:2]:                               # the users' streams stay busy beside the steps
                    with torch.cuda.stream(s):
                        junk.add_(1.0)
            if rep == 4:                                         # a collective in the middle, as a periodic gather would do
                m.pipeline_join()
                got = g.gather(box)
                g.wait()
        assert m.result_ring_frames(reset=True) == 10 * K        # raises SMK_E_SEQ if a gate timed out
        torch.cuda.synchronize()
        assert torch.equal(box, want[0]) and torch.equal(ref, want[1])
        assert torch.equal(got[0][0], want[0])
        grid, err = m.seq_status()
        assert err == 0 and grid == 256 and m.seq_recovered == 0
    finally:
        dist.destroy_process_group()
