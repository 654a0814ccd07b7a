#!/bin/bash
# round 6: the whole bench line (headline + secondary with the captured fine-tune step and evaluate()), launch tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r6_06_bench.json 2> gpurun_out/r6_06_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_06_bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "vs_baseline")})
print("roofline", {k: d["roofline"].get(k) for k in ("bound", "achieved", "peak", "frac", "traffic")}, d["roofline"].get("hbm"))
s = d.get("secondary", {})
for c in s.get("fine_tune", []):
    print({k: c.get(k) for k in ("shape", "aggregate", "ms_per_step", "ms_per_step_eager", "launch", "capture_s", "capture_error")})
for r in s.get("evaluate", {}).get("runs", []):
    print({k: r.get(k) for k in ("relation_table", "seconds", "triples_per_s", "candidate_scores_per_s", "candidate_scores_per_s_replays_only", "seconds_by_part", "in_flight", "metrics")})
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("parity", {}).get("scores_bit_equal"))
PY
tail -5 gpurun_out/r6_06_bench.err
timeout 1200 python -m pytest tests/test_launch_gpu.py -x -q 2>&1 | tail -5
