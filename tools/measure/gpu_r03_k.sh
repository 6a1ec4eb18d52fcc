#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03k
O=$R/gpurun_out/r03k
export SMK_GRAPH=1
for rep in 1 2; do
python tools/measure/gpu_step_only.py 8 2>/dev/null | grep ms/step
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/measure/gpu_step_only.py 8 2>/dev/null | grep ms/step
HIP_FORCE_DEV_KERNARG=0 python tools/measure/gpu_step_only.py 8 2>/dev/null | grep ms/step
HIP_FORCE_DEV_KERNARG=1 python tools/measure/gpu_step_only.py 8 2>/dev/null | grep ms/step
done > $O/env_ab.txt
cat $O/env_ab.txt
cd /tmp && export TMPDIR=/tmp
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/tools/measure/gpu_step_only.py 8 > $O/out.txt 2> $O/err.txt
find $O/prof -name "*kernel_trace.csv" -delete
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i "copyBuffer\|conv_seq" "$f" | cut -c1-120
