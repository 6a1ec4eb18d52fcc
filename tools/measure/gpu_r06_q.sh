#!/bin/bash
# Round 6: (1) f16x3 with a four-deep activation ring in conv_wreg (its K loops are three times as long as the fp16 context's);
# (2) B = 1: the two levers VERDICT r5 #5 names, through the knobs that exist -- shortcut convolutions on a side stream (concurrency=1:
#     fork / join as graph edges) and split-K (ksplit=1: the auto rule) -- serial steps in all arms (split-K keeps the serial step)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06q; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d.get("serial_steps") or {}
print("%-28s %9.1f frames/s  %.4f ms pipelined   serial %s ms" % (sys.argv[2], d["value"], d["ms_per_step"], s.get("ms_per_step")))
PY
}
for t in wreg_stages=0 wreg_stages=4 wreg_stages=0 wreg_stages=4; do
  timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --workload sharp_b8_f16x3 --no-cpu-baseline --no-also --no-long --tune $t > $O/x3_$t.json 2>> $O/bench.err
  line $O/x3_$t.json "f16x3 B=8 $t"
done 2>&1 | tee $O/x3_stages.txt
for t in pp=1 concurrency=1 ksplit=1 pp=1 concurrency=1 ksplit=1; do
  timeout 300 python bench.py --gpus 1 --steps 300 --warmup 30 --workload sharp_b1_f16 --no-cpu-baseline --no-also --no-long --tune $t > $O/b1_$t.json 2>> $O/bench.err
  line $O/b1_$t.json "fp16 B=1 $t"
done 2>&1 | tee $O/b1_levers.txt
tail -3 $O/bench.err
