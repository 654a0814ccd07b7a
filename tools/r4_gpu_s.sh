#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4s
mkdir -p $O
timeout 120 python -m pytest tests/test_fused_update_gpu.py -x -q -k "lds and case0 and mul" > $O/first.txt 2>&1
rc=$?
tail -5 $O/first.txt
if [ $rc -eq 124 ]; then echo "HANG in the first test: stopping"; exit 1; fi
timeout 300 python -m pytest tests/test_fused_update_gpu.py -x -q > $O/tests_fused.txt 2>&1
tail -5 $O/tests_fused.txt
timeout 200 python tools/beside_probe.py fb15k237 8 > $O/probe_fb.txt 2>&1; cat $O/probe_fb.txt
