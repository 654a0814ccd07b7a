#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4o
mkdir -p $O
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_training_gpu.py -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 600 python - > $O/train.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
for c in (lambda: sb.train_case("fb15k237"), lambda: sb.train_case("yago310"), lambda: sb.train_case("fb15k237", aggr="max")):
    print(json.dumps(c()), flush=True)
PY
cat $O/train.txt | cut -c1-250
