#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 0 8000 11000 15000; do
echo "=== ULTRA_STREAM_FILL=$v"
ULTRA_STREAM_FILL=$v timeout 200 python tools/beside_probe.py fb15k237 8 2>&1 | grep -E "hipGraph|^form 3:|per partition" | tail -3
done
bash tools/gpu_ab.sh "ULTRA_STREAM_FILL=0" "ULTRA_STREAM_FILL=11000"
