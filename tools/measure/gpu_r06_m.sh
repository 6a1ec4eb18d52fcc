#!/bin/bash
# Round 6: split-operand context: vectorised cvt_in / maxpool (parity again), tile choices for its long-K shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06m; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1800 python -m pytest tests/test_gpu_x3.py -x -q -s -k "end_to_end or 1024" 2>&1 | grep -E "f16x3|passed|failed|Error|error|assert" | tail -8 | tee $O/pytest.txt
timeout 300 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-also --no-long --workload sharp_b8_f16x3 > $O/bench.json 2>> $O/bench.err
python -c "
import json; d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('f16x3', d['value'], d['ms_per_step'])" | tee $O/bench.txt
timeout 600 python tools/measure/gpu_x3_tiles.py 8 2>&1 | grep -v amdgpu | tee $O/x3_tiles_b8.txt
