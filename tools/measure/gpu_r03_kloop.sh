#!/bin/bash
# Round 3, needs a library built with `make -C siammask_amd/csrc clean all MEASURE=1`: the K loop with parts removed -- B = 64 per-launch tiles
# (time, then effective clock per variant), a 64x128 sequence tile, the k-step issue order A/B inside conv_seq -- and the traffic-only skeleton.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r03kloop; mkdir -p $O; rm -rf $O/clk
python -c "from siammask_amd import _lib; assert _lib.tune_get('measure_build') == 1, 'build the library with MEASURE=1'" || exit 1
[ -x tools/kloop_skeleton.bin ] && timeout 200 tools/kloop_skeleton.bin > $O/skeleton.txt 2>&1
timeout 300 python tools/measure/gpu_seq_probe3.py > $O/seq_kloop_ablation.txt 2>&1; echo "probe3 exit $?"
timeout 300 python tools/measure/gpu_seq_order_ab.py > $O/seq_order_ab.txt 2>&1; echo "order exit $?"
timeout 400 python tools/measure/gpu_b64_ablate.py > $O/b64_ablate.txt 2>&1; echo "ablate exit $?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/clk -- python $R/tools/measure/gpu_b64_clock.py > $O/clk.out 2> $O/clk.err; echo "clk exit $?"
cd $R; python tools/measure/clock_stats.py $O/clk > $O/clock_stats.txt 2>&1; cat $O/clock_stats.txt
find $O/clk -name "*.csv" -size +4M -delete
