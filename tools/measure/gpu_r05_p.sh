#!/bin/bash
# pipelined steps beside the streams of a multi-rank run, under different hardware-queue limits (profiles/r05p_*)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05p; mkdir -p $out
T="tests/test_gpu_dist.py::test_pipelined_steps_beside_the_streams_of_a_multi_rank_run"
for q in default 2 3 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== GPU_MAX_HW_QUEUES=$q" >> $out/log.txt
  timeout 300 python -m pytest "$T" -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|SmkError|Error|assert" | head -8 >> $out/log.txt
done
unset GPU_MAX_HW_QUEUES
echo "== bench default" >> $out/log.txt
timeout 300 python bench.py --steps 50 --warmup 10 --no-long 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','serial_steps','fallbacks')})" >> $out/log.txt 2>&1
for q in 2 3; do
echo "== bench GPU_MAX_HW_QUEUES=$q" >> $out/log.txt
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 50 --warmup 10 --no-long 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','serial_steps','fallbacks')})" >> $out/log.txt 2>&1
done
cat $out/log.txt
