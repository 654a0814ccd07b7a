#!/bin/bash
# (one gpurun call) plain bench.py by the number of hardware queues the HIP runtime maps the process's streams onto
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'])"; }
for q in "$@"; do
GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "GPU_MAX_HW_QUEUES=$q"
done
