#!/bin/bash
# (record of a call: PROBE_CALIBRATE and ULTRA_PART_ADJUST_FILE were removed from the tree after it -- no gain, profiles/r5_experiments.txt)
# (one gpurun call, round 5) profile-guided schedule, INCREMENTAL: traced launches -> per-partition step deltas -> a few typical rows moved
# from the late partitions' streams to the early ones' (ULTRA_PART_ADJUST_FILE, plan.cpp), everything else stays where it was dealt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5u
mkdir -p $O
export TMPDIR=/tmp
{
PROBE_CALIBRATE=5 ULTRA_PART_ADJUST_FILE=$PWD/$O/adj_fb256.txt timeout 300 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
PROBE_CALIBRATE=5 PROBE_CAL_ALPHA=1.0 ULTRA_PART_ADJUST_FILE=$PWD/$O/adj_fb256_a1.txt timeout 300 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
PROBE_CALIBRATE=5 PROBE_GRID=192 ULTRA_PART_ADJUST_FILE=$PWD/$O/adj_fb192.txt timeout 300 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
PROBE_CALIBRATE=4 ULTRA_PART_ADJUST_FILE=$PWD/$O/adj_codex.txt timeout 300 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids | cut -c1-200
} 2>&1 | tee $O/calibration_incremental.txt
