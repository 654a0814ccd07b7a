#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4w
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1
tail -3 $O/tests_gpu.txt
bash tools/r4_gpu_g.sh A
