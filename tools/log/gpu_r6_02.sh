#!/bin/bash
# round 6, call 2: launch shapes of the layer-update backward (small and large row counts); the train tests again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r6_02_conv_bwd_shapes.txt
: > $out
for shape in 0,0,0 0,8,8 0,4,4 0,2,2 0,1,1 0,4,1 0,8,1 0,2,1 0,8,4 0,8,2 0,4,2; do
  ULTRA_CONV_BWD_SHAPE=$shape timeout 300 python tools/conv_bwd_probe.py >> $out 2>&1
done
# the large sizes with fewer workgroups per launch shape (two waves a SIMD is the register limit)
for shape in 128,8,8 256,6,6; do
  ULTRA_CONV_BWD_SHAPE=$shape timeout 300 python tools/conv_bwd_probe.py 116328 985456 >> $out 2>&1
done
cat $out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -3
timeout 600 python tools/train_graph_probe.py fb15k237 2>&1 | tail -1 | cut -c1-400
