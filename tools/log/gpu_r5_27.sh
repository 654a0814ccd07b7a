#!/bin/bash
# Closed-form boundary under autograd + first layer on the sources' edges: parity tests, step times, kernel tables.
OUT=gpurun_out/r5aa
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_models_gpu.py tests/test_rspmm_gpu.py tests/test_layers_gpu.py -x -q > $OUT/tests.txt 2>&1
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
for shape in fb15k237 yago310; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$shape -o run -- \
    python "$OLDPWD/tools/train_probe.py" $shape > /dev/null 2>&1)
find /tmp/prof_$shape -name "*kernel_stats.csv" -exec cp {} $OUT/${shape}_kernel_stats.csv \;
done
