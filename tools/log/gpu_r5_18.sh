#!/bin/bash
# (one gpurun call, round 5) plain bench.py: slots' stream priority x batches in flight x how many streams of that level were made before
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5r
mkdir -p $O
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'])"; }
{
for k in 2 3; do for prio in 0 -1; do for skip in 0 1 2 3; do
ULTRA_SLOT_STREAM_PRIORITY=$prio ULTRA_SLOT_STREAM_SKIP=$skip timeout 300 python bench.py --steps 20 --warmup 5 --in-flight $k --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "plain, in flight $k, priority $prio, skip $skip"
done; done; done
} 2>&1 | tee $O/plain_streams.txt
