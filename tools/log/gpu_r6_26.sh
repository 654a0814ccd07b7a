#!/bin/bash
# round 6, call 26: per-kernel averages of the captured fine-tuning step in a fresh process (a) and after a pipelined forward lived in the process (c)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_26
export TMPDIR=/tmp
for v in a c; do
  rm -rf /tmp/prof_$v
  (cd /tmp && PROBE_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o run -- python "$OLDPWD/tools/train_after_pipeline_probe.py" 2>&1 | grep "ms_per_step")
  find /tmp/prof_$v -name "*kernel_stats.csv" -exec cp {} gpurun_out/r6_26/stats_$v.csv \;
done
