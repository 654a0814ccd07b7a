#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== default"; timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== relation layer capped at 128 registers"; ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_dolcap.so timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== 12-wave order kernels + capped relation layer"; ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_w12.so timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== default"; timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== w12 parity"; ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_w12.so timeout 600 python -m pytest tests/test_order_gpu.py tests/test_fused_update_gpu.py tests/test_baseline_parity_gpu.py -x -q 2>&1 | tail -3
