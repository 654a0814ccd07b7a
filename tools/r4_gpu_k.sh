#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 1 2; do
echo "== prefill on";  timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== prefill off"; ULTRA_NO_PREFILL=1 timeout 200 python tools/step_probe.py 2>&1 | tail -2
done
