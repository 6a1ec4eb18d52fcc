#!/bin/bash
# Round 4, third GPU contact: the pair-split (2-D) form of the fused conv3 + 1x1 pairs (c3c1p_tile.inc): per-op parity, then the
# knob A/B on the whole step (in one process: off/on/off/on), phase stamps, the B = 8 end-to-end gates with it on; the B = 64
# per-kernel table (missing from call b: ring_commit rejected B = 64).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04c; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -k "pair_split or fused" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/pytest_pair.txt
timeout 300 python tools/measure/gpu_knob_ab.py seq_pair2d 8 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/ab_pair2d.txt
SMK_TUNE=seq_pair2d=1 timeout 300 python -m pytest "tests/test_gpu_e2e.py::test_bench_configuration_b8_end_to_end" tests/test_gpu_e2e.py::test_persistent_sequences_and_tail_fusion_are_what_runs_at_b8 tests/test_gpu_e2e.py::test_uneven_teams_do_not_write_over_each_other tests/test_gpu_ring.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/pytest_e2e_pair2d.txt
SMK_TUNE=seq_pair2d=1 SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep "seq clk" > $O/seqclk_pair2d.txt; grep "total\|l3.2\|l2.2" $O/seqclk_pair2d.txt | tail -12
timeout 200 python3 bench.py --workload sharp_b64_f16 --steps 40 --warmup 10 --no-also --no-cpu-baseline --no-long --profile-out $O/layers_b64.json > $O/bench_b64.json 2> $O/bench_b64.err; tail -2 $O/bench_b64.err
python - <<PY
import json
d=json.loads(open("$O/bench_b64.json").read().strip().splitlines()[-1]); print("b64", d["value"], d["ms_per_step"])
for r in d["roofline"]["kernels"][:16]: print("   ", r)
PY
