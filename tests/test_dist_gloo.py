"""N>1 path on CPU: world_size-2 gloo run of the stream sharding + end-of-batch gather
(siammask_amd/dist.py).  The data path has no collective; only fixed-size results are
gathered, and un-sharding restores global stream order."""
import os
import socket

import torch
import torch.multiprocessing as mp

from siammask_amd import dist as sdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, l, w = sdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    mine = sdist.shard_streams(n_streams, rank, world)
    s_local = (n_streams + world - 1) // world
    T = 3
    boxes = torch.zeros(s_local, T, 5)
    masks = torch.zeros(s_local, T, 7, dtype=torch.float16)
    for i, sid in enumerate(mine):            # result of stream sid at frame t encodes (sid, t)
        for t in range(T):
            boxes[i, t] = sid * 100 + t
            masks[i, t] = sid + t / 8.0
    g = sdist.ResultGather(torch.device("cpu"))
    gb, gm = g.gather(boxes, masks)
    g.wait()
    ub, um = sdist.unshard(gb, n_streams), sdist.unshard(gm, n_streams)
    ok = all(float(ub[sid, t, 0]) == sid * 100 + t and abs(float(um[sid, t, 0]) - (sid + t / 8.0)) < 1e-2
             for sid in range(n_streams) for t in range(T))
    q.put((rank, ok, tuple(ub.shape)))
    torch.distributed.destroy_process_group()


def test_gloo_world2_shard_gather_unshard():
    world, n_streams = 2, 5                    # ragged: rank 0 owns 3 streams, rank 1 owns 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n_streams, 3, 5) for _, _, shape in res)


def test_sharding_is_a_partition():
    for world in (1, 2, 4, 8):
        for n in (1, 7, 8, 64):
            owned = [s for r in range(world) for s in sdist.shard_streams(n, r, world)]
            assert sorted(owned) == list(range(n))
            assert all(sdist.stream_owner(s, world) == r for r in range(world) for s in sdist.shard_streams(n, r, world))


def test_rank_cpus_follow_the_gpus_numa_node(tmp_path):
    """8 ranks on a 2-socket node: every rank gets the CPUs of ITS GPU's NUMA node (sysfs), never an empty set; without
    NUMA information an even contiguous share of the allowed CPUs (siammask_amd/dist.py rank_cpus)."""
    sysfs = tmp_path
    for node, cpulist in ((0, "0-63,128-191"), (1, "64-127,192-255")):
        d = sysfs / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpulist + "\n")
    bdfs = ["0000:%02x:00.0" % b for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5)]
    for i, b in enumerate(bdfs):
        d = sysfs / "bus" / "pci" / "devices" / b
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % (0 if i < 4 else 1))
    allowed = set(range(256))
    for r, b in enumerate(bdfs):
        cpus, node = sdist.rank_cpus(r, 8, b, allowed, str(sysfs))
        assert node == (0 if r < 4 else 1)
        assert cpus == (set(range(0, 64)) | set(range(128, 192)) if r < 4 else set(range(64, 128)) | set(range(192, 256)))
    # a cgroup that only allows part of the node: intersected, still non-empty
    cpus, node = sdist.rank_cpus(0, 8, bdfs[0], set(range(8, 24)), str(sysfs))
    assert cpus == set(range(8, 24)) and node == 0
    # no NUMA information (numa_node = -1 / no sysfs entry): even contiguous shares that partition the allowed set
    seen = set()
    for r in range(8):
        cpus, node = sdist.rank_cpus(r, 8, "0000:aa:00.0", allowed, str(sysfs))
        assert node == -1 and len(cpus) == 32 and not (cpus & seen)
        seen |= cpus
    assert seen == allowed
    cpus, _ = sdist.rank_cpus(5, 8, None, {3}, str(sysfs))           # fewer CPUs than ranks: everybody gets what there is
    assert cpus == {3}
