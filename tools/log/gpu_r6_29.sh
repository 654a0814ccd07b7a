#!/bin/bash
# round 6, call 29: the GPU suite twice more on the final tree (flakiness), then collections A, B, C
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6
for i in 1 2; do timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -2; done > gpurun_out/r6/r6_gpu_tests_repeat.txt
cat gpurun_out/r6/r6_gpu_tests_repeat.txt
bash tools/collect_profiles.sh A r6 > /dev/null
bash tools/collect_profiles.sh B r6 > /dev/null
bash tools/collect_profiles.sh C r6 > /dev/null
ls gpurun_out/r6
