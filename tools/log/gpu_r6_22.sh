#!/bin/bash
# round 6, call 22: the shader clock under the update backward (s_memtime against s_memrealtime in workgroup 0), with and without its matrix instructions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_22
for fl in 2048 3840 2304 2560 3072; do
  echo "PROBE_FLAGS=$fl"
  PROBE_FLAGS=$fl timeout 300 python tools/conv_bwd_probe.py 116328 985456 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6_22/probe.txt
