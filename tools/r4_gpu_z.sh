#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for shape in codex_l wn18rr; do
for sh in 1 0; do
    echo "=== $shape share_chip=$sh"
    PROBE_SHAPE=$shape PROBE_SHARE=$sh timeout 280 python tools/step_probe.py 3 24 2>&1 | tail -2
done
done
