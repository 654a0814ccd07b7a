#!/bin/bash
# timing-only A/B of form 3 over variant libraries (no correctness checks: some variants compute wrong results on purpose)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4u
mkdir -p $O
for v in "" $VARIANTS; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_$v.so; fi
    echo "=== variant: ${v:-default}"
    timeout 200 python tools/beside_probe.py fb15k237 8 2>&1 | grep -v amdgpu.ids | tee $O/probe_$v.txt | grep -E "hipGraph|^form 3:|update wave" -A1 | grep -v "^--\|max over"
done
