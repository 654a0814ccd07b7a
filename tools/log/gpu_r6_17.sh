#!/bin/bash
# round 6, call 17: query rows through gather (backward = one scatter_add), tests + step time + a fresh eager timeline of the FB15k237-shaped step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_17
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_train_gpu.py tests/test_training_gpu.py tests/test_models_gpu.py tests/test_launch_gpu.py -x -q -m gpu > gpurun_out/r6_17/tests.txt 2>&1
tail -5 gpurun_out/r6_17/tests.txt
timeout 600 python tools/train_graph_probe.py fb15k237 yago310 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_17/probe.txt
rm -rf /tmp/tl_fb
(cd /tmp && PROBE_ONLY=eager timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_fb -- \
    python "$OLDPWD/tools/train_graph_probe.py" fb15k237 > /dev/null 2>&1)
python tools/train_timeline.py $(ls /tmp/tl_fb/*/*_kernel_trace.csv | head -1) > gpurun_out/r6_17/timeline_fb.txt 2>&1
