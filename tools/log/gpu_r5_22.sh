#!/bin/bash
# Fine-tune step: phase timings and a kernel timeline (start / end stamps) of the FB15k237-shape step.
OUT=gpurun_out/r5v
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/train_phases.py fb15k237 > $OUT/phases.txt 2>&1
timeout 300 python tools/train_phases.py yago310 >> $OUT/phases.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o run -- \
    python "$OLDPWD/tools/train_probe.py" > /dev/null 2>&1)
find /tmp/prof_tl -name "*kernel_trace.csv" -exec cp {} $OUT/finetune_kernel_trace.csv \;
ls -la $OUT
cat $OUT/phases.txt
