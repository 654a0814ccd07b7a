#!/bin/bash
# round 6, call 21: the update backward's f32 products as six bf16 products each (ULTRA_CONV_BWD_SPLIT bit mask): tests and per-call time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_21
S="${SPLITS:-0 1}"
for sp in $S; do
  echo "ULTRA_CONV_BWD_SPLIT=$sp"
  ULTRA_CONV_BWD_SPLIT=$sp timeout 900 python -m pytest tests/test_training_gpu.py -x -q -m gpu -k "conv_update or layer_update" 2>&1 | tail -4
  ULTRA_CONV_BWD_SPLIT=$sp timeout 300 python tools/conv_bwd_probe.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6_21/probe.txt
