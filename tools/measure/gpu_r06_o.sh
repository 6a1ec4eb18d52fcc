#!/bin/bash
# Round 6: the mask head as a persistent tile loop at large batches (smk_tune nchw_persist): parity at B = 64 / 16, then off / on / off / on
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06o; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_pipeline.py -x -q -k "b64 or 64 or 16 or 24" 2>&1 | tail -3 | tee $O/pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/bit_equal.txt
import torch, numpy as np
from siammask_amd import _lib, synth
from siammask_amd.custom import build
B = 32
m = build("sharp", dtype="f16", graph=False, max_batch=B); m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped")); m = m.eval().cuda()
z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=3)).cuda(); x = torch.from_numpy(synth.image_batch(B, 255, stream0=9)).cuda()
m.template(z)
outs = []
for v in (0, 1, 0, 1):
    _lib.tune(nchw_persist=v)
    outs.append(m.track_mask(x)[2].clone())
torch.cuda.synchronize()
print("mask logits B=32: persistent loop == one tile per workgroup:", torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3]), float(outs[0].abs().max()))
PY
for t in nchw_persist=0 nchw_persist=1 nchw_persist=0 nchw_persist=1; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-long --workload sharp_b64_f16 --tune $t > $O/b64_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/b64_$t.json").read().strip().splitlines()[-1])
print("$t", d["value"], d["ms_per_step"], [(k["kernel"], round(k["us_per_step"], 1), round(k["achieved"], 1)) for k in d["roofline"]["kernels"] if "nchw" in k["kernel"]])
PY
done 2>&1 | tee $O/b64_ab.txt
tail -3 $O/bench.err
