#!/bin/bash
# A/B on one box: the library as built vs variants (tools/build_variant.py), form 3 and the plain aggregate
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4t
mkdir -p $O
ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_guard.so timeout 90 python tools/spin_guard_probe.py > $O/guard.txt 2>&1
tail -2 $O/guard.txt
if ! grep -q "^0 reports" $O/guard.txt; then echo "a spin gave up: stopping"; exit 1; fi
timeout 300 python -m pytest tests/test_fused_update_gpu.py tests/test_order_gpu.py -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for v in "" $VARIANTS; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_$v.so; fi
    echo "=== variant: ${v:-default}"
    timeout 200 python tools/beside_probe.py fb15k237 8 2>&1 | grep -v amdgpu.ids | tee $O/probe_$v.txt | grep -E "hipGraph|form 3|update wave" -A2 | grep -v "^--"
done
