#!/bin/bash
# round 6, call 1: does the captured training step work, what does it cost, and the ordered launch list of both forms
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py -x -q > gpurun_out/r6_01_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r6_01_tests.txt
timeout 600 python tools/train_graph_probe.py fb15k237 yago310 > gpurun_out/r6_01_probe.txt 2>&1
for form in eager captured; do
  rm -rf /tmp/tl_$form
  PROBE_ONLY=$form timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$form -- python tools/train_graph_probe.py fb15k237 > /dev/null 2>&1
  f=$(ls /tmp/tl_$form/*/*_kernel_trace.csv | head -1)
  python tools/train_timeline.py $f > gpurun_out/r6_01_timeline_$form.txt 2>&1
done
tail -5 gpurun_out/r6_01_tests.txt; cat gpurun_out/r6_01_probe.txt | cut -c1-600
