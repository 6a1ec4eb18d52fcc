#!/bin/bash
# First-contact script for a fresh GPU box: environment facts + full GPU test run, everything
# logged under gpurun_out/ (only that directory comes back from gpurun).
mkdir -p gpurun_out
{
  echo "== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "Name:|Compute Unit|Max Clock" | head -20
  echo "== nproc"; nproc; lscpu | grep "Model name"
  echo "== reference present?"; ls /root/reference 2>&1 | head -3
  echo "== torch"; python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))"
  echo "== maps"; python - <<'PY'
import torch, os
from siammask_amd import _lib
L=_lib.lib()
print("version", hex(L.smk_version()))
print([l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "siammask" in l][::4])
PY
} > gpurun_out/env.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x -k "ops" > gpurun_out/pytest_ops.log 2>&1
echo "ops exit $?" >> gpurun_out/env.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -k "e2e" > gpurun_out/pytest_e2e.log 2>&1
echo "e2e exit $?" >> gpurun_out/env.log
tail -5 gpurun_out/pytest_ops.log gpurun_out/pytest_e2e.log
