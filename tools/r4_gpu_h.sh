#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in fresh one_input prereplay fresh; do timeout 120 python tools/first_run_probe.py $m 2>&1 | tail -1; done
