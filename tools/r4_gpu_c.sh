#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4c
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
cat $O/config3.txt | cut -c1-600
