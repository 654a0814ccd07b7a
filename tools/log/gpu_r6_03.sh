#!/bin/bash
# round 6, call 3: dense relation gradient + rows backward as gathers (oracle tests), the step with both, YAGO3-10's launch list
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_round6_gpu.py tests/test_train_gpu.py tests/test_training_gpu.py tests/test_models_gpu.py -x -q > gpurun_out/r6_03_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r6_03_tests.txt
tail -15 gpurun_out/r6_03_tests.txt
timeout 600 python tools/train_graph_probe.py fb15k237 yago310 > gpurun_out/r6_03_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6_03_probe.txt | cut -c1-420
for shape in fb15k237 yago310; do
  rm -rf /tmp/tl_$shape
  PROBE_ONLY=eager timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$shape -- python tools/train_graph_probe.py $shape > /dev/null 2>&1
  f=$(ls /tmp/tl_$shape/*/*_kernel_trace.csv | head -1)
  python tools/train_timeline.py $f > gpurun_out/r6_03_timeline_$shape.txt 2>&1
done
rm -rf /tmp/tl_cap
PROBE_ONLY=captured timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_cap -- python tools/train_graph_probe.py fb15k237 > /dev/null 2>&1
f=$(ls /tmp/tl_cap/*/*_kernel_trace.csv | head -1)
tail -n 1500 $f | gzip > gpurun_out/r6_03_captured_trace_tail.csv.gz
head -1 $f > gpurun_out/r6_03_captured_trace_header.txt
sed -n '/---- by kernel/,$p' gpurun_out/r6_03_timeline_yago310.txt | head -40 | cut -c1-130
