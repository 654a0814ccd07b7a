#!/bin/bash
# (one gpurun call, round 5) two-level dealing of the twelve-walker schedule: what a row costs its PARTITION (steps)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5h
mkdir -p $O
export TMPDIR=/tmp
f3() { env "$@" timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for r in 0 10 20 28 40 60; do echo "part row cost $r: $(f3 ULTRA_PART_ROW_COST_12=$r)"; done
echo "--- ... with quartet shares 1.35,1.15,0.9"
for r in 0 14 28 40; do echo "part row cost $r: $(f3 ULTRA_PART_ROW_COST_12=$r ULTRA_STREAM_SHARES_12=1.35,1.15,0.9)"; done
echo "--- ... 192 workgroups"
for r in 0 20 28 40; do echo "part row cost $r: $(f3 ULTRA_PART_ROW_COST_12=$r PROBE_GRID=192)"; done
echo "--- max aggregate"
for r in 0 20 40 80; do echo "part row cost $r: $(ULTRA_PART_ROW_COST_12=$r timeout 120 python tools/form3_probe.py fb15k237 8 max 2>&1 | grep -v amdgpu.ids | tail -1)"; done
echo "--- codex_l"
for r in 0 28 60 120 250; do echo "part row cost $r: $(ULTRA_PART_ROW_COST_12=$r timeout 120 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids | tail -1)"; done
echo "--- wn18rr (form 3 on request)"
for r in 0 28 120; do echo "part row cost $r: $(ULTRA_PART_ROW_COST_12=$r timeout 120 python tools/form3_probe.py wn18rr 8 2>&1 | grep -v amdgpu.ids | tail -1)"; done
} 2>&1 | tee $O/part_row_cost.txt
