#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4r
mkdir -p $O
timeout 600 python -m pytest tests/test_eval_gpu.py tests/test_models_gpu.py -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 600 python tools/eval_speed.py 512 2>&1 | tail -2
timeout 600 python tools/eval_speed.py 4096 2>&1 | tail -2
