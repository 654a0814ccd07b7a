#!/bin/bash
# round 6, call 20: the two-role update backward with the weights wave's operands requested two groups ahead
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_20
timeout 900 python -m pytest tests/test_training_gpu.py -x -q -m gpu > gpurun_out/r6_20/tests.txt 2>&1
tail -3 gpurun_out/r6_20/tests.txt
for f in 1 2 1 2; do
  echo "ULTRA_CONV_BWD_FUSED=$f"
  ULTRA_CONV_BWD_FUSED=$f timeout 300 python tools/conv_bwd_probe.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6_20/probe.txt
for fl in 256 512 1024 1792; do
  echo "ULTRA_CONV_BWD_FUSED=1 PROBE_FLAGS=$fl"
  ULTRA_CONV_BWD_FUSED=1 PROBE_FLAGS=$fl timeout 300 python tools/conv_bwd_probe.py 116328 985456 2>&1 | grep -v amdgpu.ids
done | tee -a gpurun_out/r6_20/probe.txt
