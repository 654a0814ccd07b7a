#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4i
mkdir -p $O
ULTRA_BENCH_UPDATE_FORM=2 ULTRA_BENCH_PMC_KEEP="$PWD/$O/pmc_beside" timeout 900 python bench.py --no-cpu-baseline --no-secondary > $O/bench_beside.json 2> $O/bench_beside.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4i/bench_beside.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "repeats")})
r = d["roofline"]
for pt in r["points"]:
    print({k: pt.get(k) for k in ("shape", "ms_per_launch", "hbm_bytes", "traffic_over_compulsory", "l2_hit_rate", "hbm_frac_measured", "l1_rate_frac")})
print(r.get("in_graph", {}).get("kernel_avg_us"))
PY
tail -2 $O/bench_beside.err
