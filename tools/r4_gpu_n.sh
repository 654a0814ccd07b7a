#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== default"; timeout 200 python tools/coresidency_probe.py 2>&1 | tail -2
echo "== w12 + capped relation layer"; ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_w12.so timeout 200 python tools/coresidency_probe.py 2>&1 | tail -2
