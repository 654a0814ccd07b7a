#!/bin/bash
# stream shares per wave quartet / stream costs: summary lines of tools/overlap_trace.py
#   [LIB=variant.so] [SHAPE=codex_l] bash tools/share_sweep.sh "q0,q1,q2,q3|chunk,row,step" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in "$@"; do
    echo "== shares=${c%%|*} costs=${c##*|} ${LIB:-}"
    ULTRA_AMD_LIB="${LIB:-}" ULTRA_STREAM_SHARES="${c%%|*}" ULTRA_STREAM_COSTS="${c##*|}" timeout 100 python tools/overlap_trace.py ${SHAPE:-fb15k237} 2>&1 | grep -E "us per call|^all|^ *[0-9]+ +[0-9]+ +[0-9]+ +[0-9]+ +[0-9]+ +[0-9]+ +[0-9]+ +[0-9]+"
done
