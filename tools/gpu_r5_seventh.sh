#!/bin/bash
# (one gpurun call, round 5) does a profile-guided rebalancing of the partitions' stream budgets converge?  tools/form3_probe.py
# PROBE_CALIBRATE: traced launch -> per-partition multipliers -> new schedule, four rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5g
mkdir -p $O
export TMPDIR=/tmp
{
PROBE_CALIBRATE=4 ULTRA_PART_WEIGHTS_FILE=$PWD/$O/w_fb256.txt timeout 300 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids
PROBE_CALIBRATE=4 PROBE_CAL_ALPHA=0.5 ULTRA_PART_WEIGHTS_FILE=$PWD/$O/w_fb256_a5.txt timeout 300 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids
PROBE_CALIBRATE=4 PROBE_GRID=192 ULTRA_PART_WEIGHTS_FILE=$PWD/$O/w_fb192.txt timeout 300 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids
PROBE_CALIBRATE=3 ULTRA_PART_WEIGHTS_FILE=$PWD/$O/w_fbmax.txt timeout 300 python tools/form3_probe.py fb15k237 8 max 2>&1 | grep -v amdgpu.ids
PROBE_CALIBRATE=3 ULTRA_PART_WEIGHTS_FILE=$PWD/$O/w_codex.txt timeout 300 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee $O/calibration.txt
