#!/bin/bash
# (one gpurun call) the first timed run of bench.py against its repeats, by the number of untimed steps in front of it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'])"; }
if [ $# -eq 0 ]; then set -- 0 16 64 128 256; fi
for n in "$@"; do
timeout 300 python bench.py --steps 20 --warmup 5 --pre-warm $n --repeats 3 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "pre-warm $n"
done
