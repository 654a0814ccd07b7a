#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4w
mkdir -p $O
timeout 600 python -m pytest tests/test_rspmm_gpu.py tests/test_models_gpu.py -x -q -k "layer0 or forward or bit" > $O/tests_l0.txt 2>&1
tail -3 $O/tests_l0.txt
echo "== step"; timeout 200 python tools/step_probe.py 2>&1 | tail -2
FORMS=0 bash tools/r4_gpu_x.sh 2>&1 | grep -A24 "== ULTRA_BENCH" | head -26
