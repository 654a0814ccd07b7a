#!/bin/bash
# (one gpurun call) A/B of the benchmark step on one box: tools/step_probe.py under the environment settings given as arguments
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "$@"; do
    echo "=== $cfg"
    env $cfg timeout 250 python tools/step_probe.py 7 40 2>&1 | tail -2
done
