#!/bin/bash
# side-by-side order kernel: schedule-cost and build-variant sweep (summary lines of tools/overlap_trace.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {  # label, lib, costs
    echo "== $1 costs=$3"
    ULTRA_AMD_LIB="$2" ULTRA_STREAM_COSTS="$3" timeout 100 python tools/overlap_trace.py 2>&1 | grep -E "us per call|^   0 |^  13 |^all|chain fit"
}
for c in "$@"; do run base "" "$c"; done
