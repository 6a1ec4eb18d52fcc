#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03t
O=$R/gpurun_out/r03t
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "bench_configuration_b8 or tight or loose or packed" 2>&1 | tail -3 > $O/pytest.txt
SMK_L1_CLK=1 timeout 300 python tools/measure/gpu_seqclk.py 2>&1 | grep "l1 clk" | tail -3 > $O/l1clk.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/tools/measure/gpu_step_only.py 8 > $O/out.txt 2> $O/err.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete
cat $O/pytest.txt $O/l1clk.txt; grep ms/step $O/out.txt | cut -c1-30; grep -i "l1_block" $O/kernel_stats.csv | cut -d, -f1-4
