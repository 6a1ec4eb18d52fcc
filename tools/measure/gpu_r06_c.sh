#!/bin/bash
# Round 6: is the B = 64 matrix work clock / power limited?  conv_pp vs conv_wreg on l3.0.ds / l3.c2 with zero, ReLU-like and uniform activations;
# then the B = 64 step with and without conv_pp (off / on / off / on)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06c; rm -rf $O; mkdir -p $O
for f in uniform relu zero; do
  echo "== activations: $f" | tee -a $O/fill_ab.txt
  SMK_BENCH_FILL=$f timeout 300 python tools/measure/gpu_pp_bench.py 64 10 3 2>&1 | grep -E "l3.c2|l3.0.ds|conv_search" | tee -a $O/fill_ab.txt
done
export SMK_GRAPH=1
for t in pp=0 pp=1 pp=0 pp=1; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-long --workload sharp_b64_f16 --tune $t > $O/b64_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/b64_$t.json").read().strip().splitlines()[-1])
print("$t", d["value"], d["ms_per_step"], [(k["kernel"], k["launches"], round(k["us_per_step"], 1), round(k["achieved"], 1)) for k in d["roofline"]["kernels"][:6]])
PY
done 2>&1 | tee $O/b64_ab.txt
tail -3 $O/bench.err
