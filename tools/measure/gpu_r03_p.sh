#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03p
O=$R/gpurun_out/r03p
export SMK_GRAPH=1
cd /tmp && export TMPDIR=/tmp
for V in 0 1 2; do
  rm -rf $O/prof$V
  SMK_TUNE=xc_full=$V timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $O/prof$V -- python $R/tools/measure/gpu_step_only.py 8 > $O/out$V.txt 2> $O/err$V.txt
  f=$(find $O/prof$V -name "*kernel_stats.csv" | head -1)
  echo "xc_full=$V : $(grep ms/step $O/out$V.txt | cut -c1-24) : $(grep -i xcorr "$f" | cut -d, -f1-4)"
  find $O/prof$V -name "*kernel_trace.csv" -delete
done
