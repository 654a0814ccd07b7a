#!/bin/bash
# round 6, call 18: the training step's readout as one autograd node (readout_train.hip): tests + step time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_18
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_train_gpu.py tests/test_training_gpu.py tests/test_models_gpu.py tests/test_launch_gpu.py -x -q -m gpu > gpurun_out/r6_18/tests.txt 2>&1
tail -15 gpurun_out/r6_18/tests.txt
timeout 600 python tools/train_graph_probe.py fb15k237 yago310 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_18/probe.txt
