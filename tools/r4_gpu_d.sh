#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4d
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -5 $O/tests_all.txt
timeout 600 python - > $O/config3.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
for aggr in ("max", "sum"):
    print(json.dumps(sb.forward_parity_case("codex_l", aggr, "ultra_50g", n_batch=1)), flush=True)
print(json.dumps(sb.forward_parity_case("fb15k237", "max", "ultra_50g", n_batch=1)), flush=True)
PY
cut -c1-330 $O/config3.txt
ULTRA_BENCH_PMC_KEEP="$PWD/$O/pmc" timeout 900 python bench.py --no-secondary > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4d/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "repeats")})
r = d["roofline"]
print({k: r.get(k) for k in ("ms_per_launch", "frac", "frac_compulsory", "l1_rate_frac", "traffic_over_compulsory", "in_graph")})
print(d.get("modes"))
print(d["parity"]["scores_bit_equal"], d["parity"]["scores"], d["parity"]["rank_mismatches"])
PY
