#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b
mkdir -p $O
for f in 0 256 512; do
  echo "== PROBE_FLAGS=$f"
  PROBE_FLAGS=$f timeout 200 python tools/beside_probe.py fb15k237 8 2>&1 | grep -E "back to back|hipGraph|form|end of work|max over"
done > $O/flags.txt 2>&1
cat $O/flags.txt
timeout 300 python -m pytest tests/test_fused_update_gpu.py tests/test_order_gpu.py -x -q > $O/tests_new.txt 2>&1
tail -5 $O/tests_new.txt
