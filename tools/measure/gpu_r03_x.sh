#!/bin/bash
# round 3: the B = 64 K loop with parts removed (time, then effective clock per variant), the k-step issue order A/B inside conv_seq
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r03x; mkdir -p $O; rm -rf $O/clk
timeout 400 python tools/measure/gpu_b64_ablate.py > $O/b64_ablate.txt 2>&1; echo "ablate exit $?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/clk -- python $R/tools/measure/gpu_b64_clock.py > $O/clk.out 2> $O/clk.err; echo "clk exit $?"
cd $R; python tools/measure/clock_stats.py $O/clk > $O/clock_stats.txt 2>&1; cat $O/clock_stats.txt
find $O/clk -name "*.csv" -size +4M -delete
