#!/bin/bash
# (record of a call: the kernel variant it probes -- relation_grad accumulated in LDS by the input-gradient walk, ULTRA_RG_PROBE -- was removed after it: profiles/r5_experiments.txt)
# One-walk backward (input_grad + relation_grad): parity tests, step times, kernel table.
OUT=gpurun_out/r5y
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_rspmm_gpu.py tests/test_models_gpu.py -x -q > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 900 python - > $OUT/finetune.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
for shape in ("fb15k237", "yago310"):
    print(json.dumps(sb.train_case(shape)), flush=True)
PY
cat $OUT/finetune.txt
for shape in fb15k237 yago310; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$shape -o run -- \
    python "$OLDPWD/tools/train_probe.py" $shape > /dev/null 2>&1)
find /tmp/prof_$shape -name "*kernel_stats.csv" -exec cp {} $OUT/${shape}_kernel_stats.csv \;
head -12 $OUT/${shape}_kernel_stats.csv | cut -c1-130
done
