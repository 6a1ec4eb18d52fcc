#!/bin/bash
# Round 4: Refine chain -- the nearest-upsampling table look-ups of post0 / post1 hoisted out of the k-steps (six independent reads per
# tile instead of two dependent ones per k-step).  Product library against the variant with the old look-ups (conv_igemm.hip holds
# chain_mask_kernel), alternating processes; the per-layer stamps of the stand-alone chain; parity of the product library.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04aa; mkdir -p $O
export SMK_GRAPH=1
B="python3 bench.py --steps 400 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2 3 4; do
  for arm in product chainold; do
    unset SMK_LIB; [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
    timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], [r['us_per_step'] for r in d['roofline']['kernels'] if r['kernel']=='chain_mask'])" | tee -a $O/ab.txt
  done
done
unset SMK_LIB
for arm in product; do
  unset SMK_LIB; [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
  echo "== $arm" | tee -a $O/chain_layers.txt
  SMK_GRAPH=0 SMK_CHAIN_CLK=1 timeout 120 python - <<'PY' 2>&1 | grep "refine_chain layers" | tail -2 | tee -a $O/chain_layers.txt
import torch
from siammask_amd import synth
from siammask_amd.custom import build
B = 8
m = build("sharp", dtype="f16", max_batch=B, graph=False)
m.load_state_dict(synth.torch_state_dict("sharp", "synthetic_damped"))
m = m.eval().cuda()
z = torch.from_numpy(synth.smooth_image_batch(B, 127, stream0=3)).cuda()
x = torch.from_numpy(synth.smooth_image_batch(B, 255, stream0=3)).cuda()
m.template(z)
for i in range(3):
    m.track_mask(x)
    m.track_refine((12, 12))
torch.cuda.synchronize()
PY
done
unset SMK_LIB
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "refine or chain or bench_configuration or fp16_tight" 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/pytest.txt
