#!/bin/bash
# round 6, call 32: where the slow 20-step loops come from (per-step completion events and host issue times over 300 loops, three processes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_32
for i in 1 2 3; do timeout 600 python tools/loop_jitter_probe.py 2>&1 | grep -v amdgpu.ids | tail -12; echo; done | tee gpurun_out/r6_32/out.txt
