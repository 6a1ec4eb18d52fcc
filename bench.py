#!/usr/bin/env python
"""bench.py -- frames/sec of the SiamMask per-frame inference path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch of synthetic frames per GPU:
    track_mask(search[B,3,255,255]) -> (cls, loc, 63x63 mask logits) ; track_refine(pos[B,2])
against the cached 127x127 template (template() is untimed, it runs once per stream).
Default workload = BASELINE.json configs[2] at N=1 / configs[3] at N>1: SiamMask-sharp with the
Refine module (config_davis.json shapes), B=8 streams per GPU in lock-step, fp16 storage with
fp32 accumulation, inputs resident in HBM.  N > 1 shards independent streams over GPUs (weak
scaling, no data-path collective) and gathers boxes/masks with one RCCL all_gather at the end
of the K frames, inside the timed region.

Prints ONE JSON line (driver contract) with two extra objects:
  roofline     -- dominant kernel family (the MFMA implicit-GEMM convolutions: conv_igemm_kernel and
                  its 3x3 patch-sharing sibling conv3x3_halo_kernel): algorithmic FLOPs of its
                  launches / their HIP-event durations, measured live on the launch stream by the
                  library's per-launch profiler (smk_profile);
  cpu_baseline -- the CPU port of the reference op sequence (oracle/torch_port.py) timed on
                  this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

from siammask_amd import spec, synth  # noqa: E402
from siammask_amd import dist as sdist  # noqa: E402

WORKLOADS = {
    # name: (variant, batch per GPU, dtype, refine)
    "sharp_b8_f16": ("sharp", 8, "f16", True),      # BASELINE configs[2] / configs[3]
    "sharp_b1_f16": ("sharp", 1, "f16", True),
    "sharp_b64_f16": ("sharp", 64, "f16", True),    # configs[4] regime (per GPU)
    "sharp_b8_f32": ("sharp", 8, "f32", True),
    "base_b1_f32": ("base", 1, "f32", False),       # BASELINE configs[1]
    "rpn_b1_f32": ("rpn", 1, "f32", False),         # BASELINE configs[0] shape on the GPU
}
PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3}        # dense MFMA peaks, MI355X_MICROARCH.md


def make_model(variant, dtype, batch, device):
    from siammask_amd.custom import build
    m = build(variant, dtype=dtype, max_batch=batch, graph=True)
    m.load_state_dict(synth.torch_state_dict(variant, "synthetic_damped"))
    return m.eval().to(device)


class Workload(object):
    def __init__(self, name, device, rank, n_inputs=4, batch=None, fused=True):
        self.variant, self.B, self.dtype, self.refine = WORKLOADS[name]
        if batch:
            self.B = batch
        self.name, self.device = name, device
        B = self.B
        self.model = make_model(self.variant, self.dtype, B, device)
        s0 = rank * B
        self.z = torch.from_numpy(synth.image_batch(B, 127, stream0=s0)).to(device)
        # a small ring of distinct search batches (random data, not zeros: DVFS hygiene)
        self.xs = [torch.from_numpy(synth.image_batch(B, 255, stream0=s0 + 1000 * (i + 1))).to(device)
                   for i in range(n_inputs)]
        g = np.random.Generator(np.random.PCG64(99 + rank))
        self.pos = torch.from_numpy(g.integers(8, 17, size=(B, 2)).astype(np.int32)).to(device)
        # target size in crop pixels per stream (what siamese_track derives from its state)
        self.twh = torch.from_numpy(g.uniform(40.0, 110.0, size=(B, 2)).astype(np.float32)).to(device)
        self.model.template(self.z)
        self.fused = fused
        self.last = None

    def step(self, i):
        """One frame per stream.  fused: track_mask -> on-device decode (tools/test.py:205-254) ->
        track_refine at the decoded positions, one captured graph, no host round trip.
        unfused: the reference's call sequence with a fixed refine position."""
        m = self.model
        x = self.xs[i % len(self.xs)]
        if self.fused:
            # inputs are pre-staged persistent buffers: read in place (no staging copy per frame)
            o = m.track_step(x, self.twh, refine=self.refine, stage=False)
            self.last = (o["box"], o["loc"], o["mask"], o["refine"])
        elif self.variant == "rpn":
            cls, loc = m.track(x)
            self.last = (cls, loc, None, None)
        else:
            cls, loc, mask = m.track_mask(x)
            ref = m.track_refine(self.pos) if self.refine else None
            self.last = (cls, loc, mask, ref)
        return self.last

    def gflop_per_frame(self):
        return spec.GFLOP_PER_FRAME[self.variant]


def prewarm(w, seconds):
    """Untimed: run steps until the GPU has been busy for `seconds` of wall time, so that a
    fresh box (idle clocks) does not bias the W warm-up + K timed steps that follow."""
    if seconds <= 0:
        return
    t0, i = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            w.step(i)
            i += 1
        torch.cuda.synchronize(w.device)


def timed_run(w, steps, warmup, world, gather):
    dev = w.device
    res_masks = res_box = None
    if w.refine:
        res_masks = torch.empty((steps, w.B, spec.REFINE_OUT ** 2), dtype=torch.float16, device=dev)
    # per stream and frame: the decoded box (cx, cy, w, h, score, penalty, pscore, best_id) and the
    # 127x127 refine mask logits -- the fixed-size results a tracker keeps (tools/test.py:296-311)
    # frame-major [T, B, ...]: each step writes one contiguous row (5 us less per step than [B, T, ...] slices,
    # profiles/r01_v6_keep_probe.txt)
    res_box = torch.empty((steps, w.B, 8 if w.fused else 30 * 625), dtype=torch.float32 if w.fused else torch.float16,
                          device=dev)
    for i in range(warmup):
        w.step(i)
    torch.cuda.synchronize(dev)
    if world > 1:
        # untimed: one gather of the same tensors, so that RCCL's lazily built channels / registered buffers for this
        # message size exist before the clock starts (the timed region then contains exactly one steady-state gather)
        outs = gather.gather(res_box, res_masks) if res_masks is not None else gather.gather(res_box)
        gather.wait()
        del outs
        torch.cuda.synchronize(dev)
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        cls, loc, mask, ref = w.step(i)
        # results kept for the end-of-batch gather (scores/boxes + mask logits)
        if w.fused:
            res_box[i].copy_(cls)                     # `cls` slot carries the decoded box [B,8]
        else:
            res_box[i, :, :10 * 625].copy_(cls.reshape(w.B, -1))
            res_box[i, :, 10 * 625:].copy_(loc.reshape(w.B, -1))
        if ref is not None:
            res_masks[i].copy_(ref)
    ev1.record()
    t_enq = time.perf_counter() - t0          # host time to enqueue the K steps (no device wait)
    if world > 1:
        outs = gather.gather(res_box, res_masks) if res_masks is not None else gather.gather(res_box)
        gather.wait()
        del outs
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    w.last_timing = {"host_enqueue_ms_per_step": round(t_enq / steps * 1e3, 4),
                     "gpu_event_ms_per_step": round(ev0.elapsed_time(ev1) / steps, 4)}
    return dt


CONV_FAMILY = "conv_igemm+conv3x3_halo"


def roofline(w, steps=3):
    """Per-launch HIP-event timing of every kernel (library profiler, eager launches on the
    current stream) -> achieved TFLOP/s of the dominant kernel family."""
    m = w.model
    m.profile(True)
    for i in range(steps):
        w.step(i)
    recs = m.profile_dump()
    m.profile(False)
    fam = {}
    for r in recs:
        k = r["kernel"].split("<")[0]
        k = CONV_FAMILY if k in ("conv_igemm", "conv3x3_halo") else k     # the two MFMA implicit-GEMM conv kernels
        f = fam.setdefault(k, {"ms": 0.0, "flop": 0.0, "bytes": 0.0, "calls": 0})
        f["ms"] += r["ms"]; f["flop"] += r["flop"]; f["bytes"] += r["bytes"]; f["calls"] += r["calls"]
    total_ms = sum(f["ms"] for f in fam.values())
    dom = max(fam, key=lambda k: fam[k]["ms"])
    d = fam[dom]
    peak = PEAK_TFLOPS[w.dtype]
    achieved = d["flop"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    heavy = max((r for r in recs if r["kernel"].startswith(("conv_igemm", "conv3x3_halo"))), key=lambda r: r["ms"])
    xc = fam.get("dw_xcorr")
    out = {
        "bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None,
        "launches_per_step": d["calls"] // steps,
        "avg_launch_us": round(d["ms"] * 1e3 / max(1, d["calls"]), 2),
        "share_of_gpu_time": round(d["ms"] / total_ms, 4) if total_ms else None,
        "heaviest_launch": {"id": heavy["id"], "kernel": heavy["kernel"],
                            "us": round(heavy["ms"] * 1e3 / heavy["calls"], 2),
                            "tflops": round(heavy["flop"] / (heavy["ms"] * 1e-3) / 1e12, 2)},
        "kernel_ms_per_step": round(total_ms / steps, 4),
    }
    # HBM bytes per launch from the PMC counters (collected offline by tools/measure/gpu_pmc.sh with rocprofv3
    # --pmc in separate passes and committed under profiles/; cannot be sampled from inside this process)
    pmc = os.path.join(REPO, "profiles", "pmc_traffic_%s.json" % w.name)
    if os.path.exists(pmc):
        try:
            t = json.load(open(pmc))
            out["traffic"] = t["conv_igemm_family"]["hbm_bytes_per_launch_corrected"]
            out["traffic_note"] = "bytes per launch of the conv kernels (conv_igemm, conv3x3_halo); %s; %s" % (t["source"], t["correction"])
        except Exception:  # noqa: BLE001
            pass
    out["algorithmic_bytes_per_launch"] = int(d["bytes"] / max(1, d["calls"]))
    if xc and xc["ms"] > 0:
        out["dw_xcorr"] = {"bound": "hbm", "achieved_GBps": round(xc["bytes"] / (xc["ms"] * 1e-3) / 1e9, 1),
                           "peak_GBps": 8000.0, "us": round(xc["ms"] * 1e3 / xc["calls"], 2)}
    return out, recs


def cpu_baseline(budget_s=12.0):
    """CPU port of the reference op sequence (fp32, torch CPU ops on the host cores):
    sharp track_mask + track_refine at B=1.  A short scan picks the best thread count (ATen's
    convolutions stop scaling long before 128+ cores), then as many frames as fit in ~budget_s."""
    from oracle.torch_port import TorchPort
    t = TorchPort(synth.state_dict("sharp", "synthetic_damped"), "sharp")
    z = torch.from_numpy(synth.image_batch(1, 127, stream0=0))
    x = torch.from_numpy(synth.image_batch(1, 255, stream0=1000))
    ncpu = os.cpu_count() or 1
    best_thr, best_fps = torch.get_num_threads(), 0.0
    with torch.no_grad():
        t.template(z)
        for thr in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
            torch.set_num_threads(thr)
            t.track_mask(x); t.track_refine((12, 12))
            t0 = time.perf_counter()
            for _ in range(2):
                t.track_mask(x); t.track_refine((12, 12))
            f = 2 / (time.perf_counter() - t0)
            if f > best_fps:
                best_thr, best_fps = thr, f
        torch.set_num_threads(best_thr)
        t.track_mask(x); t.track_refine((12, 12))
        n, t0 = 0, time.perf_counter()
        while True:
            t.track_mask(x); t.track_refine((12, 12))
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 400:
                break
    return {"value": round(n / el, 2), "unit": "frames/sec", "cores": best_thr, "kind": "port",
            "sample": "%d frames of sharp track_mask+track_refine, B=1, fp32, torch CPU ops (oracle/torch_port.py), "
                      "%.1f s with %d threads (best of a 8..128 scan) on %s logical CPUs" % (n, el, best_thr, ncpu)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="sharp_b8_f16", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads (B=1 fp32, ...)")
    ap.add_argument("--prewarm-seconds", type=float, default=2.0,
                    help="untimed clock/cache warm-up before the W warm-up steps")
    ap.add_argument("--unfused", action="store_true",
                    help="time the reference call sequence (track_mask, track_refine(fixed pos)) instead of the "
                         "fused device-resident step")
    ap.add_argument("--inputs", type=int, default=4, help="ring of distinct pre-staged search batches")
    ap.add_argument("--tune", default="", help="library tuning knobs, e.g. xcd_mode=0,force_tile=1")
    ap.add_argument("--profile-out", default="", help="write the per-layer launch profile (JSON) here")
    args = ap.parse_args()

    if args.tune:
        from siammask_amd import _lib
        _lib.tune(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune.split(",")})
    rank, local, world = sdist.init_from_env("nccl")
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    gather = sdist.ResultGather(dev)

    w = Workload(args.workload, dev, rank, n_inputs=args.inputs, batch=args.batch, fused=not args.unfused)
    prewarm(w, args.prewarm_seconds)
    dt = timed_run(w, args.steps, args.warmup, world, gather)
    main_timing = dict(w.last_timing)
    frames = w.B * world * args.steps
    fps = frames / dt

    try:
        roof, recs = roofline(w)
    except Exception as e:  # the profiler pass must never cost the contract line its `value`
        roof, recs = {"error": str(e)[:200]}, []
    if args.profile_out and rank == 0:
        with open(args.profile_out, "w") as f:
            json.dump({"workload": args.workload, "batch": w.B, "dtype": w.dtype, "layers": recs}, f, indent=1)

    also = {}
    if rank == 0 and world == 1 and not args.no_also:
        for name in ("base_b1_f32", "sharp_b1_f16", "sharp_b64_f16"):
            if name == args.workload:
                continue
            try:
                w2 = Workload(name, dev, 0)
                prewarm(w2, 0.5)
                k2 = max(10, min(args.steps, 100 if w2.B < 64 else 30))
                d2 = timed_run(w2, k2, 5, 1, gather)
                r2, _ = roofline(w2, 2)
                also[name] = {"fps": round(w2.B * k2 / d2, 1), "ms_per_step": round(d2 / k2 * 1e3, 3),
                              "mfma_frac": r2["frac"], "conv_tflops": r2["achieved"]}
                del w2
                torch.cuda.empty_cache()
            except Exception as e:  # secondary numbers must never kill the contract line
                also[name] = {"error": str(e)[:200]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:  # noqa: BLE001 -- a reported baseline, never a reason to lose the line
            cpu = {"error": str(e)[:200]}

    if rank == 0:
        line = {
            "metric": "frames/sec (255x255 search, 127x127 template)",
            "value": round(fps, 2), "unit": "frames/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": w.dtype, "data": "synthetic",
            "config": {"workload": "siammask_%s track_mask%s, B=%d streams per GPU in lock-step, cached template; "
                                   "BASELINE configs[%s]" % (w.variant, "+track_refine" if w.refine else "", w.B,
                                                            "2" if world == 1 else "3"),
                       "name": args.workload, "variant": w.variant, "batch_per_gpu": w.B,
                       "global_batch": w.B * world, "parallelism": "streams sharded x%d" % world,
                       "weights": "synthetic_damped (calibrated random init)", "graph": True,
                       "step": ("track_mask -> device decode -> track_refine, one graph" if w.fused
                                else "track_mask ; track_refine(fixed pos)"),
                       "gflop_per_frame": w.gflop_per_frame()},
            "roofline": roof,
            "cpu_baseline": cpu,
            "tflops_end_to_end": round(fps / world * w.gflop_per_frame() / 1e3, 2),
            "timing": main_timing,
            "also": also or None,
        }
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
