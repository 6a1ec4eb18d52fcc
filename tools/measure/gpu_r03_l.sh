#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03l
O=$R/gpurun_out/r03l
export SMK_GRAPH=1
cd /tmp && export TMPDIR=/tmp
i=0
for E in "X=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "DEBUG_HIP_KERNARG_COPY_OPT=1" "DEBUG_CLR_BLIT_KERNARG_OPT=0" "DEBUG_CLR_BLIT_KERNARG_OPT=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=64" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "ROC_USE_FGS_KERNARG=0"; do
  i=$((i+1)); rm -rf $O/prof$i
  env $E timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $O/prof$i -- python $R/tools/measure/gpu_step_only.py 8 > $O/out$i.txt 2> $O/err$i.txt
  f=$(find $O/prof$i -name "*kernel_stats.csv" | head -1)
  echo "$E : $(grep ms/step $O/out$i.txt | cut -c1-24) : $(grep -i copyBuffer "$f" | cut -d, -f1-4)"
  find $O/prof$i -name "*kernel_trace.csv" -delete
done
