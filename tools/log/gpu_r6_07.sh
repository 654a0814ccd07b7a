#!/bin/bash
# round 6: relation projection node (tests, step time), timeline with names
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_round6_gpu.py tests/test_train_gpu.py tests/test_models_gpu.py tests/test_training_gpu.py -x -q > gpurun_out/r6_07_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r6_07_tests.txt
tail -12 gpurun_out/r6_07_tests.txt
timeout 600 python tools/train_graph_probe.py fb15k237 yago310 > gpurun_out/r6_07_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6_07_probe.txt | cut -c1-420
for shape in fb15k237; do
  rm -rf /tmp/tl_$shape
  PROBE_ONLY=eager timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$shape -- python tools/train_graph_probe.py $shape > /dev/null 2>&1
  f=$(ls /tmp/tl_$shape/*/*_kernel_trace.csv | head -1)
  python tools/train_timeline.py $f > gpurun_out/r6_07_timeline_$shape.txt 2>&1
  sed -n '/---- by kernel/,$p' gpurun_out/r6_07_timeline_$shape.txt | head -60 | cut -c1-140
done
