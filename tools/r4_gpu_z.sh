#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 6 7.5 8.4 9.5; do
    echo "=== ULTRA_STREAM_STEP_12=$v"
    ULTRA_STREAM_STEP_12=$v timeout 200 python tools/beside_probe.py fb15k237 8 2>&1 | grep -E "hipGraph|^form 3:|per partition"
done
echo "=== codex_l"
for v in 6 8.4; do
ULTRA_STREAM_STEP_12=$v timeout 250 python tools/beside_probe.py codex_l 8 2>&1 | grep -E "hipGraph|^form 3:|per partition"
done
