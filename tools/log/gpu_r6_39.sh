#!/bin/bash
# round 6, call 39 (second form): the keep vector as a zero relation row in LDS (FwdParams::keep_zero) against the multiply (ULTRA_KEEP_ZERO_ROW=0): the walk alone with the
# vector tagged (one permutation, as in the step), and the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_39
for z in 1 0 1 0; do
  echo "ULTRA_KEEP_ZERO_ROW=$z"
  ULTRA_KEEP_ZERO_ROW=$z timeout 600 python tools/walk_kind_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-125
done | tee gpurun_out/r6_39/out.txt
