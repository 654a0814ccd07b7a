#!/bin/bash
# (one gpurun call, round 5) MORE workgroups than CUs for the entity layers of the batches in flight: finer-grained launches that
# interleave with the other batch's short launches as CUs come free (each workgroup still owns a CU while it runs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5q
mkdir -p $O
export TMPDIR=/tmp
{
for g in 192 256 320 384 512 768; do
    echo "shared grid $g: $(ULTRA_SHARED_GRID=$g timeout 300 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu.ids | tail -1)"
done
for g in 384 512; do
    echo "three in flight, grid $g: $(PROBE_DEPTH=3 ULTRA_SHARED_GRID=$g timeout 300 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu.ids | tail -1)"
done
for g in 256 384 512; do echo "one layer alone, grid $g: $(PROBE_GRID=$g timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170)"; done
} 2>&1 | tee $O/oversubscribed.txt
