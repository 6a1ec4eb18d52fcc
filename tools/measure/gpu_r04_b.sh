#!/bin/bash
# Round 4, second GPU contact: result ring + c3c1 residual-in-registers (RESREG): parity, then A/B of each in separate processes
# on the same box (ABAB), phase stamps, and the per-layer profiles of B = 64 / B = 1 that the next steps are planned on.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04b; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_seq.py "tests/test_gpu_e2e.py::test_bench_configuration_b8_end_to_end" tests/test_gpu_e2e.py::test_persistent_sequences_and_tail_fusion_are_what_runs_at_b8 -x -q 2>&1 | grep -v amdgpu.ids | tail -6 > $O/pytest.txt; cat $O/pytest.txt
B="python3 bench.py --steps 300 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2; do
  for arm in ring noring resreg0; do
    case $arm in
      ring) X="";;
      noring) X="--no-ring";;
      resreg0) export SMK_LIB=$R/build_variants/resreg0/libsiammask_hip.so; X="";;
    esac
    timeout 120 $B $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'], 'launches', d['roofline']['launches_per_step_all_kernels'])" | tee -a $O/ab.txt
    unset SMK_LIB
  done
done
SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep "l3.2.c3\|l2.2.c3\|total" | tee $O/seqclk_resreg1.txt
SMK_LIB=$R/build_variants/resreg0/libsiammask_hip.so SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep "l3.2.c3\|l2.2.c3\|total" | tee $O/seqclk_resreg0.txt
for wl in sharp_b64_f16 sharp_b1_f16; do
  timeout 200 python3 bench.py --workload $wl --steps 60 --warmup 10 --no-also --no-cpu-baseline --no-long --profile-out $O/layers_$wl.json > $O/bench_$wl.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/bench_$wl.json").read().strip().splitlines()[-1]); print("$wl", d["value"], d["ms_per_step"])
for r in d["roofline"]["kernels"][:14]: print("   ", r)
PY
done
