#!/bin/bash
# round 6, call 31: the training walks on slices of the batch (working set of the gathers against the last-level cache)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_31
for s in yago310 fb15k237; do timeout 600 python tools/walk_split_probe.py $s 2>&1 | grep -v amdgpu.ids | tail -6; done | tee gpurun_out/r6_31/out.txt
