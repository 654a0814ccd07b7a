#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m pytest tests/test_eval_gpu.py -x -q 2>&1 | tail -2
timeout 300 python tools/eval_speed.py 4096 2>&1 | grep evaluate
timeout 300 python tools/eval_speed.py 512 2>&1 | grep evaluate
