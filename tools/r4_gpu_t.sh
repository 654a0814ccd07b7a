#!/bin/bash
# form-3 A/B on one box: the library as built, update waves at priority 0, update waves that skip the multiplication (timing only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4t
mkdir -p $O
for v in "" prio0 skip; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_$v.so; fi
    echo "=== variant: ${v:-default}"
    PROBE_FORMS=3 timeout 200 python tools/beside_probe.py fb15k237 8 2>&1 | grep -v amdgpu.ids | tee $O/probe_$v.txt | grep -E "hipGraph|form 3|end of work|update wave"
done
