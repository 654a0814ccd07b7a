#!/bin/bash
# round 6: the update backward as ONE launch (rows waves + weights waves, dz through LDS) against the two launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_training_gpu.py tests/test_train_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -4
for f in 0 1; do
  echo "ULTRA_CONV_BWD_FUSED=$f"
  ULTRA_CONV_BWD_FUSED=$f timeout 300 python tools/conv_bwd_probe.py 2>&1 | grep -v amdgpu
  ULTRA_CONV_BWD_FUSED=$f timeout 600 python tools/train_graph_probe.py fb15k237 yago310 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['shape'], 'eager', round(d['eager_ms'], 3), 'captured', round(d['captured_ms'], 3), 'issue', round(d.get('captured_host_issue_ms', 0), 3), d['captured_loss'][:3])"
done
