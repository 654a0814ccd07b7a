#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5l
timeout 300 python tools/cumask_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5l/cumask_probe.txt
