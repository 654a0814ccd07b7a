#!/bin/bash
# (record of a call: the two-workgroup build of the update backward it measures was not kept: profiles/r5_experiments.txt)
# Update backward with two workgroups per CU (compiler held to 128 VGPRs: spills) -- step times and the YAGO3-10 kernel table.
OUT=gpurun_out/r5x
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_training_gpu.py -x -q -k "conv_update" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 900 python - > $OUT/finetune.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
for shape in ("fb15k237", "yago310"):
    print(json.dumps(sb.train_case(shape)), flush=True)
PY
cat $OUT/finetune.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -o run -- \
    python "$OLDPWD/tools/train_probe.py" yago310 > /dev/null 2>&1)
find /tmp/prof_y -name "*kernel_stats.csv" -exec cp {} $OUT/yago_kernel_stats.csv \;
head -25 $OUT/yago_kernel_stats.csv | cut -c1-150
