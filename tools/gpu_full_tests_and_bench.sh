#!/bin/bash
# (one gpurun call: /usr/local/graft/bin/gpurun --timeout 900 -- "bash tools/gpu_full_tests_and_bench.sh"; writes under gpurun_out/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4w
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1
tail -3 $O/tests_gpu.txt
bash tools/gpu_collect.sh A
