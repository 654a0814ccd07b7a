#!/bin/bash
# the first timed run of bench.py against its repeats: with the roofline block in front, without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'])"; }
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "no roofline      "
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc 2>/dev/null | pr "roofline, no pmc "
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "no roofline      "
timeout 300 python bench.py --steps 20 --warmup 50 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "warmup 50        "
