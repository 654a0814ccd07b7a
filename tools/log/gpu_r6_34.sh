#!/bin/bash
# round 6, call 34: soak -- 3000 captured AdamW steps against 3000 eager ones (FB15k237 shape; first call: 400), 300 at YAGO3-10's: parameters bit for bit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_34
{ timeout 900 python tools/train_soak_probe.py 3000 fb15k237; timeout 900 python tools/train_soak_probe.py 300 yago310; } 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r6_34/out.txt
