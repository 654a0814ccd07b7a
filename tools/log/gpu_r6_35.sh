#!/bin/bash
# round 6, call 35: soak of the pipelined forward -- 2,000 batches three in flight against one at a time, every score bit for bit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_35
timeout 900 python tools/forward_soak_probe.py 2000 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r6_35/out.txt
