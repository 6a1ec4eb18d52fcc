#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03r
O=$R/gpurun_out/r03r
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "bench_configuration_b8 or tight or loose" 2>&1 | tail -3 > $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/tools/measure/gpu_step_only.py 8 > $O/out.txt 2> $O/err.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete
cat $O/pytest.txt; grep ms/step $O/out.txt | cut -c1-30; grep -i "l1_block\|stem_pool\|conv_seq" $O/kernel_stats.csv | cut -d, -f1-4
