#!/bin/bash
# round 6, call 23: issue priority of the rows waves over the weights waves in the update backward
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_23
timeout 900 python -m pytest tests/test_training_gpu.py -x -q -m gpu -k "conv_update or layer_update" 2>&1 | tail -3
for fl in 2048 6144 2048 6144; do
  echo "PROBE_FLAGS=$fl"
  PROBE_FLAGS=$fl timeout 300 python tools/conv_bwd_probe.py 2056 3792 116328 985456 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6_23/probe.txt
