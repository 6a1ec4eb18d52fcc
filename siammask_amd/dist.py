"""Multi-GPU: independent video streams sharded over the GPUs of one node, one process per GPU.

The path shards by stream: each (video, object) stream is independent, frames
within a stream are sequential.  Every rank holds a full weight replica and tracks its own
streams in lock-step batches; there is NO data-path collective.  The only exchange is one
RCCL all_gather (torch.distributed backend 'nccl' == RCCL on ROCm) of the fixed-size results
(boxes/scores and mask logits) at the end of a batch of frames, issued on a side stream so
that it can overlap the next batch of frames.  With the 'gloo' backend the same code runs on
CPU tensors (world_size-2 tests)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun) if WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


# ---- host placement of a rank (one process per GPU): CPU affinity next to the GPU's NUMA node --------------------------------
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(pci_bus_id, sysfs="/sys"):
    """NUMA node of a PCI device ('0000:c1:00.0') from sysfs, or -1 when the platform does not say."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def rank_cpus(local_rank, n_local, pci_bus_id=None, allowed=None, sysfs="/sys"):
    """The CPUs a rank should run on.  The reference's scale-out is one process per GPU too (experiments/siammask_sharp/
    test_all.sh:68,77) and leaves placement to the OS; on an 8-GPU node with 2 sockets that lets a rank's host thread -- which
    enqueues a graph every 0.6 ms -- run a socket away from its GPU.  Rule: ALL the CPUs of the GPU's NUMA node (sysfs) that the
    process is allowed on -- ranks that share a node share its CPUs (a rank has one busy host thread; the OS spreads them inside
    the node); without NUMA information an even contiguous share of the allowed CPUs, the remainder going to the last rank.
    Never returns an empty set."""
    allowed = set(allowed if allowed is not None else os.sched_getaffinity(0))
    node = gpu_numa_node(pci_bus_id, sysfs) if pci_bus_id else -1
    if node >= 0:
        try:
            with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
                cpus = _parse_cpulist(f.read()) & allowed
            if cpus:
                return cpus, node
        except OSError:
            pass
    order = sorted(allowed)
    n = max(1, n_local)
    share = max(1, len(order) // n)
    r = local_rank % n
    lo = r * share
    hi = len(order) if r == n - 1 else lo + share          # (the tail CPUs of an uneven split belong to the last rank)
    return set(order[lo:hi]) or set(order), -1


def pin_rank(local_rank, n_local, device_index=None):
    """Apply rank_cpus() to the CALLING THREAD (os.sched_setaffinity(0, ...) is per thread on Linux; SMK_NO_AFFINITY=1 leaves
    placement alone).  Threads created afterwards inherit it; helper threads the HIP runtime / RCCL started before the call keep
    theirs -- call this before the first CUDA call where that matters (bench.py pins right after torch.cuda.set_device, i.e. the
    runtime's own early threads are not covered; the report's "pinned" says exactly: the enqueueing thread is).
    -> dict for the dry-run report."""
    bdf = None
    if device_index is not None and torch.cuda.is_available():
        try:
            p = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        except Exception:  # noqa: BLE001
            bdf = None
    cpus, node = rank_cpus(local_rank, n_local, bdf)
    pinned = False
    if os.environ.get("SMK_NO_AFFINITY", "0") != "1":
        try:
            os.sched_setaffinity(0, cpus)
            pinned = True
        except OSError:
            pinned = False
    return {"pci": bdf, "numa_node": node, "cpus": len(cpus), "cpu_first": min(cpus), "cpu_last": max(cpus), "pinned": pinned,
            "pinned_scope": "calling thread + threads created after the call"}


def shard_streams(n_streams, rank, world):
    """Round-robin assignment of stream ids to ranks -> list of global stream ids of `rank`."""
    return list(range(rank, n_streams, world))


def stream_owner(stream_id, world):
    return stream_id % world


class ResultGather(object):
    """all_gather of per-rank results [S_local, T, ...] -> [world, S_local, T, ...] on every rank.

    Every rank must contribute the same shapes (pad the last shard).  On GPUs the collective
    runs on a dedicated side stream; ``wait()`` makes the current stream wait for it.

    The inputs must be PRIVATE result buffers (bench.py's Results rows), not the views `track_step` returns: those
    alias the persistent graph I/O buffers, which the next frame overwrites while the side-stream gather still reads.
    With pipelined frame steps (Custom.set_pipeline) call ``model.pipeline_join()`` on the current stream first: the last
    frame's logits row is written by the pipeline's side stream, which torch does not know about."""

    def __init__(self, device=None, always_collective=False):
        self.device = device
        # world 1 normally needs no exchange; always_collective sends it through the backend anyway (a 1-rank RCCL all_gather
        # on the side stream): the overlap of the gather with the next batch of frames can then be exercised on ONE GPU
        self.always_collective = always_collective
        self.side = torch.cuda.Stream(device=device) if (device is not None and device.type == "cuda") else None
        self._pending = []

    def gather(self, *tensors):
        world = dist.get_world_size() if dist.is_initialized() else 1
        outs = []
        if world == 1 and not (self.always_collective and dist.is_initialized()):
            return [t.unsqueeze(0) for t in tensors]
        if self.side is not None:
            cur = torch.cuda.current_stream(self.device)
            # outputs are allocated on the CALLER's stream (that is where they are consumed after wait() and where the
            # caching allocator may recycle them); the side stream's use of them is declared with record_stream
            ins = [t.contiguous() for t in tensors]
            outs = [torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in ins]
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                for t, out in zip(ins, outs):
                    dist.all_gather_into_tensor(out, t)
                    t.record_stream(self.side)
                    out.record_stream(self.side)
        else:
            for t in tensors:
                t = t.contiguous()
                parts = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(parts, t)
                outs.append(torch.stack(parts))
        return outs

    def wait(self):
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)


def unshard(gathered, n_streams):
    """[world, S_local, ...] gathered with round-robin sharding -> [n_streams, ...] in stream order."""
    world, s_local = gathered.shape[0], gathered.shape[1]
    flat = gathered.transpose(0, 1).reshape((world * s_local,) + tuple(gathered.shape[2:]))
    return flat[:n_streams]
