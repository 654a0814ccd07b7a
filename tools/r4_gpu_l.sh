#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4l
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -4 $O/tests_all.txt
timeout 200 python tools/step_probe.py 2>&1 | tail -2
timeout 200 python tools/step_probe.py 2>&1 | tail -2
