#!/bin/bash
# Round 6: resident trunk in conv_seq_kernel (smk_tune seq_yres): parity (bit-equal to the unmarked lists), the sequence / e2e / pipeline suites,
# then the driver command off / on / off / on and the phase clocks
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06h; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_seq.py -x -q -k "resident" -s 2>&1 | grep -E "resident trunk|passed|failed|Error|error|assert" | tail -12 | tee $O/pytest_resident.txt
timeout 1500 python -m pytest tests/test_gpu_seq.py tests/test_gpu_e2e.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for t in seq_yres=0 seq_yres=1 seq_yres=0 seq_yres=1; do
  timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-long --tune $t > $O/b8_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/b8_$t.json").read().strip().splitlines()[-1])
ks = {k["kernel"]: round(k["us_per_step"], 1) for k in d["roofline"]["kernels"]}
print("$t", d["value"], d["ms_per_step"], "conv_seq", ks.get("conv_seq"), "frac", d["roofline"]["frac"])
PY
done 2>&1 | tee $O/b8_ab.txt
SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 300 python bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-long --serial 2>&1 | grep "seq clk" | tail -70 > $O/seq_phase_clocks.txt
tail -3 $O/bench.err
