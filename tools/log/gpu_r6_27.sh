#!/bin/bash
# round 6, call 27: the sampler's stream chosen by trying (tasks.overlapping_stream): the captured step fresh (a) and after a pipelined forward (c, d); tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_27
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_train_gpu.py tests/test_round6_gpu.py tests/test_launch_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in a c d b a c; do PROBE_VARIANT=$v timeout 300 python tools/train_after_pipeline_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; done | tee gpurun_out/r6_27/probe.txt
