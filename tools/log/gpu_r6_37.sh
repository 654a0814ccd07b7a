#!/bin/bash
# round 6, call 37: the gradient parity test (HIP autograd path vs CPU autograd of the oracle model in fp32 and fp64) on the FB15k237- and YAGO3-10-shaped graphs, one query x 257
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_37
free -g | head -2
for shp in fb15k237 yago310; do echo $shp; ULTRA_GRAD_PARITY_SHAPE=$shp timeout 1500 python -m pytest tests/test_models_gpu.py -q -m gpu -s -k "test_training_step_gradients_match_cpu_autograd" 2>&1 | grep -v amdgpu.ids | tail -5 ; done | tee gpurun_out/r6_37/out.txt
