#!/bin/bash
# (record of a call: the kernel variant it probes -- relation_grad accumulated in LDS by the input-gradient walk, ULTRA_RG_PROBE -- was removed after it: profiles/r5_experiments.txt)
# What the one-walk backward's LDS adds cost: ds_add_f32 against a plain store and a racy read-add-store (timing only).
OUT=gpurun_out/r5z
mkdir -p $OUT
export TMPDIR=/tmp
for probe in 0 2 4; do
(cd /tmp && ULTRA_RG_PROBE=$probe timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$probe -o run -- \
    python "$OLDPWD/tools/train_probe.py" fb15k237 > /dev/null 2>&1)
echo "probe $probe" >> $OUT/rg_probe.txt
find /tmp/prof_$probe -name "*kernel_stats.csv" -exec grep "rspmm_fwd_kernel" {} \; | cut -c1-120 >> $OUT/rg_probe.txt
done
cat $OUT/rg_probe.txt
