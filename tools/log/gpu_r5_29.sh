#!/bin/bash
# Tagged keep mask (one permutation per plan and step): parity tests and step times.
OUT=gpurun_out/r5ac
mkdir -p $OUT
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_models_gpu.py tests/test_rspmm_gpu.py -x -q > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 900 python - > $OUT/finetune.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
for shape in ("fb15k237", "yago310"):
    print(json.dumps(sb.train_case(shape)), flush=True)
print(json.dumps(sb.train_case("fb15k237", aggr="max")), flush=True)
PY
cat $OUT/finetune.txt
