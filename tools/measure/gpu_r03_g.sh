#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03g
O=gpurun_out/r03g
export SMK_GRAPH=1
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k xcorr 2>&1 | tail -3 > $O/pytest_xcorr.txt
timeout 300 python tools/measure/gpu_knob_ab.py xc_full 8,64 0,1 > $O/ab_xc_full.txt 2>&1
timeout 400 python bench.py --steps 100 --warmup 10 --no-also --no-cpu-baseline --profile-out $O/layers_b8.json > $O/bench.txt 2>&1
tail -n 2 $O/pytest_xcorr.txt; grep ms/step $O/ab_xc_full.txt; python -c "
import json
d=json.load(open('$O/layers_b8.json'))
for r in d:
    if r['id'] in ('dw_xcorr','conv_search','head0'): print(r['id'], round(r['ms']*1e3/r['calls'],2),'us')
"
