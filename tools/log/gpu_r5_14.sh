#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5n
{
PROBE_DUMP_PARTS=1 timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids
PROBE_DUMP_PARTS=1 PROBE_GRID=192 timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5n/dumps.txt 2>&1
grep -v "^PARTS" gpurun_out/r5n/dumps.txt
