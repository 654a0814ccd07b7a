#!/bin/bash
# (one gpurun call, round 5) checkpoint: GPU suite, form-3 probes on every shape, the step, and collection A (bench line + profiles)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5j
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
{
timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
PROBE_GRID=192 timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py fb15k237 8 max 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py codex_l 8 max 2>&1 | grep -v amdgpu.ids | tail -1
} 2>&1 | tee $O/form3.txt
timeout 300 python tools/step_probe.py 9 40 2>&1 | grep -v amdgpu.ids | tee $O/step.txt
bash tools/collect_profiles.sh A r5 > $O/collect_A.log 2>&1
tail -3 $O/collect_A.log
python - <<'PY'
import json
for f in ("gpurun_out/r5/r5_bench.json", "gpurun_out/r5/r5_bench_torchrun1.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "ms_per_step", round(d["ms_per_step"], 4), "median", round(d.get("ms_per_step_median", 0), 4), "repeats", d["repeats"]["ms_per_step"],
              "slots", d["config"].get("slot_streams"))
        if "roofline" in d:
            r = d["roofline"]
            print("  roofline frac", round(r["frac"], 3), "frac_compulsory", round(r["frac_compulsory"], 3), "binding", r["binding_roof"]["name"], round(r["binding_roof"]["frac"], 3),
                  "ms_per_launch", r["ms_per_launch"], "in_graph", (r.get("in_graph") or {}).get("kernel_avg_us"))
        if "modes" in d:
            print("  modes", {k: round(v.get("ms_per_step", 0), 4) for k, v in d["modes"].items()})
        if "secondary" in d:
            print("  forward", [(c["shape"], c["aggregate"], round(c["ms_per_forward"], 3)) for c in d["secondary"]["forward"]])
            print("  fine_tune", [(c.get("shape"), c.get("aggregate"), round(c.get("ms_per_step", 0), 2)) for c in d["secondary"]["fine_tune"]])
        if "parity" in d:
            print("  parity bit-equal", d["parity"]["scores_bit_equal"], "/", d["parity"]["scores"], "rank mismatches", d["parity"]["rank_mismatches"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
