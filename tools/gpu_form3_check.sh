#!/bin/bash
# (one gpurun call: /usr/local/graft/bin/gpurun --timeout 900 -- "bash tools/gpu_form3_check.sh"; writes under gpurun_out/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4s
mkdir -p $O
# hang guard first: a build whose update waves give up on an endless spin and say where (tools/spin_guard_probe.py)
ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_guard.so timeout 90 python tools/spin_guard_probe.py > $O/guard.txt 2>&1
tail -4 $O/guard.txt
if ! grep -q "^0 reports" $O/guard.txt; then echo "a spin gave up: stopping"; exit 1; fi
timeout 100 python -m pytest tests/test_fused_update_gpu.py -x -q -k "lds and (case1 or case5)" > $O/first.txt 2>&1
rc=$?
tail -3 $O/first.txt
if [ $rc -ne 0 ]; then echo "first tests failed (rc $rc): stopping"; exit 1; fi
timeout 300 python -m pytest tests/test_fused_update_gpu.py -x -q > $O/tests_fused.txt 2>&1
tail -3 $O/tests_fused.txt
timeout 200 python tools/beside_probe.py fb15k237 8 > $O/probe_fb.txt 2>&1; cat $O/probe_fb.txt
