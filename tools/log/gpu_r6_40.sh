#!/bin/bash
# round 6, call 40: the keep vector announced as such (ultra_rspmm_weight_keep) -> dropped edges as a zero relation row in LDS (FwdParams::keep_zero) against the multiply
# (ULTRA_KEEP_ZERO_ROW=0): tests, the walk alone, the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_40
timeout 1200 python -m pytest tests/test_training_gpu.py tests/test_train_gpu.py tests/test_round6_gpu.py tests/test_models_gpu.py tests/test_rspmm_gpu.py tests/test_launch_gpu.py -x -q -m gpu 2>&1 | tail -3
for z in 1 0 1 0; do
  echo "ULTRA_KEEP_ZERO_ROW=$z"
  ULTRA_KEEP_ZERO_ROW=$z timeout 600 python tools/walk_kind_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-90
  ULTRA_KEEP_ZERO_ROW=$z PROBE_ONLY=captured timeout 600 python tools/train_graph_probe.py fb15k237 yago310 2>&1 | grep -v amdgpu.ids | cut -c1-105
done | tee gpurun_out/r6_40/out.txt
