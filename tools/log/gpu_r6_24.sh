#!/bin/bash
# round 6, call 24: the full GPU suite on HEAD
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6
git rev-parse HEAD > /dev/null 2>&1
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r6/r6_gpu_tests.txt
cat gpurun_out/r6/r6_gpu_tests.txt
