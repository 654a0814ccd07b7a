#!/bin/bash
# round 6, call 19: the update backward with three waves a SIMD (conv_update_bwd_three_kernel) against the two-role launch: tests, per-call time, step time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_19
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_train_gpu.py tests/test_round6_gpu.py -x -q -m gpu > gpurun_out/r6_19/tests.txt 2>&1
tail -15 gpurun_out/r6_19/tests.txt
for f in 1 2 0 2 1; do
  echo "ULTRA_CONV_BWD_FUSED=$f"
  ULTRA_CONV_BWD_FUSED=$f timeout 300 python tools/conv_bwd_probe.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6_19/probe.txt
for f in 1 2; do
  echo "ULTRA_CONV_BWD_FUSED=$f"
  ULTRA_CONV_BWD_FUSED=$f PROBE_ONLY=captured timeout 600 python tools/train_graph_probe.py fb15k237 yago310 2>&1 | grep -v amdgpu.ids | cut -c1-400
done | tee gpurun_out/r6_19/step.txt
