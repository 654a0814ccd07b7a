#!/bin/bash
# round 6: the forward as two captures, relations half on a high-priority stream (ULTRA_SPLIT_PRIORITY=1), three in flight
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for split in 0 1 0 1; do
  echo "ULTRA_SPLIT_PRIORITY=$split"
  ULTRA_SPLIT_PRIORITY=$split PROBE_DEPTH=3 timeout 600 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu | tail -3
done
for split in 0 1; do
  rm -rf /tmp/pt_$split
  ULTRA_SPLIT_PRIORITY=$split PROBE_DEPTH=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_$split -- python tools/step_probe.py 3 60 > /dev/null 2>&1
  f=$(ls /tmp/pt_$split/*/*_kernel_trace.csv | head -1)
  python - $f <<'PY' > gpurun_out/r6_11_pipeline_split$split.txt
import csv, sys
sys.path.insert(0, "tools")
import pipeline_timeline as pt
rows = list(csv.DictReader(open(sys.argv[1])))
end = max(int(r["End_Timestamp"]) for r in rows)
# a window in the middle of the last measured run: drop the last 8 ms (drain), keep 3 ms
keep = [r for r in rows if int(r["End_Timestamp"]) < end - 8_000_000]
import tempfile, os
w = csv.DictWriter(open("/tmp/mid.csv", "w", newline=""), fieldnames=list(rows[0].keys()))
w.writeheader(); w.writerows(keep)
pt.main("/tmp/mid.csv", 3.0, False)
PY
  echo "split $split:"; cat gpurun_out/r6_11_pipeline_split$split.txt
done
