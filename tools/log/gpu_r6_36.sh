#!/bin/bash
# round 6, call 36: the BASELINE-size parity test of the headline configuration on 32 batches (512 rankings, 64 oracle forwards) instead of 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_36
ULTRA_PARITY_BATCHES=32 timeout 1500 python -m pytest tests/test_baseline_parity_gpu.py -q -m gpu -s -k "config2 or config1" 2>&1 | grep -v amdgpu.ids | grep "config\|passed\|failed" | tee gpurun_out/r6_36/out.txt
