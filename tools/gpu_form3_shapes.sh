#!/bin/bash
# the one-launch layer's forms over shapes and batch sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4v
mkdir -p $O
for cfg in "fb15k237 8" "fb15k237 16" "fb15k237 4" "wn18rr 8" "codex_l 8" "fb15k237 8 max"; do
    echo "=== $cfg"
    timeout 250 python tools/beside_probe.py $cfg 2>&1 | grep -v amdgpu.ids | tee "$O/probe_${cfg// /_}.txt" | grep -E "== two|hipGraph|^form 3:|update wave" -A1 | grep -v "^--\|max over"
done
