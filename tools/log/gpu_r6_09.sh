#!/bin/bash
# round 6: relation-graph layer forms in the pipelined step: 0 = 160 registers, 1 = four 256-thread workgroups a CU, 2 = one 1024-thread workgroup
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_order_gpu.py -x -q -k "two_row" 2>&1 | tail -3
for lean in 0 2 1 2 0 2; do
  echo "ULTRA_DOL_LEAN=$lean"
  ULTRA_DOL_LEAN=$lean timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-roofline --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms_per_step', round(d['ms_per_step'], 4), d['repeats']['ms_per_step'], 'table', d.get('modes', {}).get('relation_table', {}).get('ms_per_step'), d['config'].get('slot_streams', {}).get('settle'))
"
done
