#!/bin/bash
# (one gpurun call: /usr/local/graft/bin/gpurun --timeout 900 -- "bash tools/gpu_collect.sh"; writes under gpurun_out/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
part="${1:-A}"
bash tools/collect_profiles.sh $part r4 > /dev/null 2>&1
ls -la gpurun_out/r4 | head -30
if [ "$part" = "A" ]; then
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/r4_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "repeats")})
r = d["roofline"]
print({k: r.get(k) for k in ("achieved", "frac", "traffic", "frac_compulsory", "traffic_over_compulsory", "l1_rate_frac")})
print("in_graph", r.get("in_graph", {}).get("kernel_avg_us"))
for pt in r.get("points", []):
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in pt.items() if k in ("shape", "ms_per_launch", "hbm_bytes", "compulsory_bytes", "l2_hit_rate", "traffic_over_compulsory", "ms_per_launch_aggregate_only")})
print([ (c["shape"], c["aggregate"], round(c["ms_per_forward"],3), c["parity"]["bit_equal"]) for c in d["secondary"]["forward"]])
print([ (c["shape"], c["aggregate"], round(c["ms_per_step"],2)) for c in d["secondary"]["fine_tune"]])
print(d["modes"]["one_batch_in_flight"]["ms_per_step"], d["modes"]["two_launch_layers"]["ms_per_step"])
PY
tail -3 gpurun_out/r4/bench.err
fi
