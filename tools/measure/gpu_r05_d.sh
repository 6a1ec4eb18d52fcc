#!/bin/bash
# Round 5: 96 x 256 conv_wreg tile for conv_search (A/B), hipStreamWaitValue32 probe, fresh per-kernel tables (B = 8, 64, 1), e2e parity
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05d; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 60 ./tools/order_probe.out > $O/order_probe.txt 2>&1; grep "2b" $O/order_probe.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also"
for t in wreg96=1 wreg96=0; do
  timeout 300 python bench.py $B --serial --tune $t > $O/bench_serial_$t.json 2>> $O/bench.err
  timeout 300 python bench.py $B --tune $t > $O/bench_pipe_$t.json 2>> $O/bench.err
done
timeout 300 python bench.py $B --serial --workload sharp_b64_f16 --steps 20 --profile-out $O/layers_b64.json > $O/bench_b64.json 2>> $O/bench.err
timeout 300 python bench.py $B --serial --workload sharp_b1_f16 --steps 50 --profile-out $O/layers_b1.json > $O/bench_b1.json 2>> $O/bench.err
python - <<PY
import json
for n in ("bench_serial_wreg96=1", "bench_serial_wreg96=0", "bench_pipe_wreg96=1", "bench_pipe_wreg96=0", "bench_b64", "bench_b1"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], "200:", d.get("value_200_steps"), "serial:", (d.get("serial_steps") or {}).get("ms_per_step"))
        if n in ("bench_serial_wreg96=1", "bench_b64", "bench_b1"):
            for x in d["roofline"]["kernels"]: print("   ", x)
    except Exception as e:
        print(n, "ERR", e)
PY
tail -3 $O/bench.err
