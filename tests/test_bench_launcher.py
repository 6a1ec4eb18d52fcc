"""`python bench.py --gpus N` must create its own N ranks when no launcher did (the reference scales out
by process-per-GPU sharding, experiments/siammask_sharp/test_all.sh:68,77).  Exercised here on CPU: gloo backend, world
size 2, StubWorkload (result shapes of the fused step, no kernels) -- the launcher, rendezvous, world-size assertion,
timed-loop protocol and end-of-batch gather are the real code."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, cwd=REPO,
                          capture_output=True, text=True, timeout=timeout)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out          # exactly ONE line, from rank 0
    return json.loads(lines[0])


def test_bare_gpus2_spawns_two_ranks():
    r = _run(["--stub", "--gpus", "2", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["steps"] == 4
    assert line["scaling"] == "weak" and line["config"]["name"] == "stub"


def test_world_size_mismatch_fails_loudly():
    # launched as a single rank of a 1-rank job but asked for 2: must not silently measure one device
    r = _run(["--stub", "--gpus", "2", "--steps", "2", "--warmup", "1"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])          # real workload, no devices here
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)
    r = _run(["--steps", "2", "--warmup", "1"])
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_single_rank_stub_line():
    r = _run(["--stub", "--steps", "3", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_dry_run_reports_every_rank_and_its_host_placement():
    """`--gpus 2 --dry-run`: no measurement; rank 0 prints ONE line with both ranks' reports, each with a
    non-empty CPU set (siammask_amd/dist.py pin_rank) -- what a first multi-GPU run could fail on is visible before it is timed"""
    r = _run(["--stub", "--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["dry_run"] and line["n_gpus"] == 2 and len(line["ranks"]) == 2
    assert sorted(x["rank"] for x in line["ranks"]) == [0, 1]
    assert len({x["pid"] for x in line["ranks"]}) == 2
    for x in line["ranks"]:
        assert x["host_placement"]["cpus"] >= 1
