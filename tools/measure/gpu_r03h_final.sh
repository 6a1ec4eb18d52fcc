#!/bin/bash
# Round 3, second half (fused conv3 + conv1 pairs, patch-sharing 3x3 tiles inside conv_seq_kernel): full GPU suite; PMC passes
# (separate runs, kernel-trace only) and the traffic summary tied to the kernel sources (sha256) that bench.py checks -- FIRST, so
# that the bench lines below carry roofline.traffic; effective clock per kernel; the driver's exact bench command (+ per-layer
# profile); the 200-step line; rocprofv3 kernel stats of the driver's command; per-layer stamps of the sequence.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03hfinal
O=$R/gpurun_out/r03hfinal
export SMK_GRAPH=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
bash tools/measure/gpu_pmc.sh \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" 2>&1 | tail -40 > $O/pmc_tail.txt
cp gpurun_out/pmc/pmc_by_kernel.json $O/pmc_by_kernel.json
python tools/measure/clock_stats.py gpurun_out/pmc/pass2 > $O/kernel_clocks.txt 2>&1     # GRBM_GUI_ACTIVE / dispatch duration (sum over 8 XCDs: MHz / 8)
python tools/measure/pmc_traffic.py $O/pmc_by_kernel.json sharp_b8_f16 "profiles/r03h_pmc_by_kernel.json" > $O/pmc_traffic_sharp_b8_f16.json
cp $O/pmc_traffic_sharp_b8_f16.json profiles/pmc_traffic_sharp_b8_f16.json               # (on the box: what bench.py reads)
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --profile-out $O/layers_b8.json > $O/bench_driver_cmd.json 2>/dev/null; echo "driver-cmd bench exit $?"
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-also --no-cpu-baseline > $O/bench_200steps.json 2>/dev/null; echo "200-step bench exit $?"
SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py > $O/seqclk.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/bench.py --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also > $O/rocprof_bench.json 2> $O/rocprof.err
echo "rocprof exit $?"
find $O/prof -name "*kernel_trace.csv" -delete
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/rocprofv3_kernel_stats.csv
rm -rf $O/prof
cd $R
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("driver cmd:", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "traffic", d["roofline"]["traffic"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["config"]["persistent_sequences"])
print("also:", {k: (v.get("fps"), v.get("ms_per_step"), v.get("mfma_frac")) for k, v in d.get("also", {}).items()})
d=json.loads(open("$O/bench_200steps.json").read().strip().splitlines()[-1]); print("200 steps:", d["value"], d["ms_per_step"], "traffic", d["roofline"]["traffic"])
PY
head -12 $O/rocprofv3_kernel_stats.csv | cut -c1-150
