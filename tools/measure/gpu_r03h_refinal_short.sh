#!/bin/bash
# Round 3, last GPU minutes: after the buffer-layout fix (stage-private intermediates) -- the uneven-team check, the PMC passes +
# traffic summary tied to the new kernel sources, one bench line.  (The full suite of this state runs at round end.)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r03hshort; mkdir -p $O
export SMK_GRAPH=1
timeout 60 python tools/measure/gpu_b12_sweep.py 10,12 2>&1 | grep -v amdgpu.ids | tee $O/uneven.txt
bash tools/measure/gpu_pmc.sh \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" 2>&1 | tail -12 > $O/pmc_tail.txt
cp gpurun_out/pmc/pmc_by_kernel.json $O/pmc_by_kernel.json
python tools/measure/pmc_traffic.py $O/pmc_by_kernel.json sharp_b8_f16 "profiles/r03h_pmc_by_kernel.json" > $O/pmc_traffic_sharp_b8_f16.json
cp $O/pmc_traffic_sharp_b8_f16.json profiles/pmc_traffic_sharp_b8_f16.json
timeout 60 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline > $O/bench_driver_cmd_noalso.json 2>/dev/null; echo "bench exit $?"
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd_noalso.json").read().strip().splitlines()[-1])
print("bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "traffic", d["roofline"]["traffic"], d["config"]["persistent_sequences"])
PY
