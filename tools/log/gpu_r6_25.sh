#!/bin/bash
# round 6, call 25: the captured fine-tuning step after what bench.py's process has done before it (streams, a pipelined forward)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_25
for v in a b c d a c; do PROBE_VARIANT=$v timeout 300 python tools/train_after_pipeline_probe.py 2>&1 | grep -v amdgpu.ids | tail -2; done | tee gpurun_out/r6_25/probe.txt
