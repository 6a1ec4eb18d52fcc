#!/bin/bash
# Round 5: the PMC passes of the final script again, on SERIAL steps (see gpu_pmc.sh) -> profiles/pmc_traffic_sharp_b8_f16.json, r05_pmc_by_kernel.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05pmc; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
python tools/measure/src_hash.py > $O/kernel_sources_sha256.txt
bash tools/measure/gpu_pmc.sh \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" 2>&1 | tail -30 > $O/pmc_tail.txt
cp gpurun_out/pmc/pmc_by_kernel.json $O/pmc_by_kernel.json
python tools/measure/pmc_traffic.py $O/pmc_by_kernel.json sharp_b8_f16 "profiles/r05_pmc_by_kernel.json" > $O/pmc_traffic_sharp_b8_f16.json
python -c "
import json; t=json.load(open('$O/pmc_traffic_sharp_b8_f16.json')); print(t['by_kernel'].get('conv_seq_kernel')); print({k[:50]: (v['mfma_util_est'], v['hbm_bytes_per_launch_corrected'], v['dispatches']) for k, v in t['per_instantiation'].items() if 'seq' in k})"
