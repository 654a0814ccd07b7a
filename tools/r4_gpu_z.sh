#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "" $VARIANTS; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_$v.so; fi
    echo "=== variant: ${v:-default}"
    timeout 120 python tools/chain_probe.py fb15k237 8 2>&1 | grep -E "staged|chains done|least"
    timeout 200 python tools/beside_probe.py fb15k237 8 2>&1 | grep -E "hipGraph|== two"
done
