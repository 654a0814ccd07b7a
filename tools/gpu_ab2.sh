#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_ab.sh "ULTRA_CHAIN_LIMIT_FACTOR=1.9" "ULTRA_CHAIN_LIMIT_FACTOR=2.3"
timeout 400 python tools/share_probe.py 8 256 192 160 2>&1 | grep grid
