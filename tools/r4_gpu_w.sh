#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4w
mkdir -p $O
ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_guard.so timeout 90 python tools/spin_guard_probe.py > $O/guard.txt 2>&1
tail -2 $O/guard.txt
if ! grep -q "^0 reports" $O/guard.txt; then echo "a spin gave up: stopping"; exit 1; fi
echo "== library's choice"; timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== tail form";        PROBE_UPDATE_FORM=1 timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== library's choice"; timeout 200 python tools/step_probe.py 2>&1 | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1
tail -5 $O/tests_gpu.txt
