#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4j
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -4 $O/tests_all.txt
ULTRA_BENCH_PMC_KEEP="$PWD/$O/pmc" timeout 900 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json, csv
d = json.loads(open("gpurun_out/r4j/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "repeats")})
rows=list(csv.DictReader(open('gpurun_out/r4j/pmc/bench_kernel_stats_inflight1.csv')))
for r in rows[:14]:
    print("%-80s %5s %9.2f %6s" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
tail -2 $O/bench.err
