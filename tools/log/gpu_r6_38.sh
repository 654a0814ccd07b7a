#!/bin/bash
# round 6, call 38: the training forward walk on the re-associating plan against the reference-order stream walk, with and without the keep vector
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_38
timeout 900 python tools/walk_kind_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r6_38/out.txt
