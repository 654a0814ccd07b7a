#!/bin/bash
# First-layer backward as one kernel: parity tests, step times, FB15k237 timeline.
OUT=gpurun_out/r5ab
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_models_gpu.py -x -q > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 900 python - > $OUT/finetune.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
for shape in ("fb15k237", "yago310"):
    print(json.dumps(sb.train_case(shape)), flush=True)
PY
cat $OUT/finetune.txt
timeout 300 python tools/train_phases.py fb15k237 > $OUT/phases.txt 2>&1; cat $OUT/phases.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o run -- \
    python "$OLDPWD/tools/train_probe.py" > /dev/null 2>&1)
find /tmp/prof_tl -name "*kernel_trace.csv" -exec cp {} $OUT/finetune_kernel_trace.csv \;
