#!/bin/bash
# (one gpurun call, round 5) the step with two batches in flight by the workgroups of the aggregation launches, and three in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5k
mkdir -p $O
export TMPDIR=/tmp
{
for g in 192 176 160 184 208 224 144 128; do
    echo "shared grid $g: $(ULTRA_SHARED_GRID=$g timeout 300 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
done
echo "whole-chip launches: $(PROBE_SHARE=0 timeout 300 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
for g in 192 176 160; do
    echo "three in flight, shared grid $g: $(PROBE_DEPTH=3 ULTRA_SHARED_GRID=$g timeout 300 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
done
} 2>&1 | tee $O/shared_grid.txt
