#!/bin/bash
# Round 5: full GPU suite on the restructured library (wash arms behind MEASURE=1, ABI 1.5) + pipe_sig A/B + argmax vs fp64 oracle
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05e; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
cp gpurun_out/argmax_agreement.json $O/ 2>/dev/null
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also"
for t in pipe_sig=1 pipe_sig=0 pipe_sig=1,pipe_eager=2; do
  timeout 300 python bench.py $B --tune $t > $O/bench_$t.json 2>> $O/bench.err
done
python - <<PY
import json
for n in ("bench_pipe_sig=1", "bench_pipe_sig=0", "bench_pipe_sig=1,pipe_eager=2"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], "200:", d.get("value_200_steps"), "lat:", d.get("latency"), "serial:", (d.get("serial_steps") or {}).get("ms_per_step"))
    except Exception as e:
        print(n, "ERR", e)
PY
python -c "
import json; d=json.load(open('$O/argmax_agreement.json')); print(d.get('vs_fp64_oracle')); print(d['summary'])"
tail -3 $O/bench.err
