#!/bin/bash
# Round 6: conv_search as the persistent sequence's last record (smk_tune seq_search): parity suites, A/B off / on / off / on; why is (l3.5.c3, adjust) not a
# fused pair (SMK_SEQ_DEBUG); phase clocks with the resident trunk; GPU_MAX_HW_QUEUES=1 falls back to serial steps
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06i; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1800 python -m pytest tests/test_gpu_seq.py tests/test_gpu_e2e.py tests/test_gpu_pipeline.py tests/test_gpu_dropin.py tests/test_gpu_argmax.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for t in seq_search=0 seq_search=1 seq_search=0 seq_search=1; do
  timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-long --tune $t > $O/b8_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/b8_$t.json").read().strip().splitlines()[-1])
ks = {k["kernel"]: round(k["us_per_step"], 1) for k in d["roofline"]["kernels"]}
print("$t", d["value"], d["ms_per_step"], "conv_seq", ks.get("conv_seq"), "frac", d["roofline"]["frac"], {k: v for k, v in ks.items() if "wreg" in k})
PY
done 2>&1 | tee $O/b8_ab.txt
SMK_GRAPH=0 SMK_SEQ_DEBUG=1 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep -E "seq clk|seq fuse" > $O/seq_phase_clocks.txt
tail -75 $O/seq_phase_clocks.txt | cut -c1-250
GPU_MAX_HW_QUEUES=1 timeout 120 python - <<'PY' 2>&1 | tail -4 | tee $O/hw_queues_1.txt
import time, torch, numpy as np
from siammask_amd import synth
from siammask_amd.custom import build
B = 8
m = build("sharp", dtype="f16", graph=True, max_batch=B)
m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped")); m = m.eval().cuda()
z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=3)).cuda(); x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=3)).cuda()
twh = torch.full((B, 2), 70.0, dtype=torch.float64).cuda()
m.template(z); m.set_pipeline(1)
t0 = time.time()
for i in range(20): o = m.track_step(x, twh, refine=True, stage=False)
m.pipeline_join(); torch.cuda.synchronize()
print("GPU_MAX_HW_QUEUES=1: 20 steps with set_pipeline(1) in %.3f s, seq_status %s" % (time.time() - t0, m.seq_status()))
PY
tail -3 $O/bench.err
