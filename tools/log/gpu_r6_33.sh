#!/bin/bash
# round 6, call 33: the rows forward with four edges in flight per group: oracle tests, the step, a timeline's kernel time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_33
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_training_gpu.py tests/test_train_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -3
PROBE_ONLY=captured timeout 600 python tools/train_graph_probe.py fb15k237 yago310 2>&1 | grep -v amdgpu.ids | cut -c1-330
for shape in fb15k237 yago310; do
  rm -rf /tmp/tl_$shape
  (cd /tmp && PROBE_ONLY=eager timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl_$shape -o run -- python "$OLDPWD/tools/train_graph_probe.py" $shape > /dev/null 2>&1)
  grep "rspmm_rows_kernel" $(find /tmp/tl_$shape -name "*kernel_stats.csv" | head -1) | cut -c1-200
done
