#!/bin/bash
# round 6: the lean relation-graph layer (four workgroups a CU) A/B in the pipelined step; new golden sampler test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_order_gpu.py -x -q -k "dense_order or two_row" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -k "sampler or projection" 2>&1 | tail -3
for lean in 0 1 0 1; do
  echo "ULTRA_DOL_LEAN=$lean"
  ULTRA_DOL_LEAN=$lean timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms_per_step', round(d['ms_per_step'], 4), 'repeats', d.get('repeats'), 'relation_table', d.get('modes', {}).get('relation_table', {}).get('ms_per_step'))
"
done
