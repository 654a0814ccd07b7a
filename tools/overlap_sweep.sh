#!/bin/bash
# side-by-side order kernel (measurement build): schedule-cost sweep, summary lines of tools/overlap_trace.py
#   bash tools/overlap_sweep.sh LIB "chunk,row,step,side,hub_share" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIB="$1"; shift
for c in "$@"; do
    echo "== costs=$c"
    ULTRA_AMD_LIB="$LIB" ULTRA_STREAM_COSTS="$c" timeout 100 python tools/overlap_trace.py 2>&1 | grep -E "us per call|^   [0-3] |^  13 |^all|chain fit"
done
