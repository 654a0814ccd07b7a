#!/bin/bash
# (one gpurun call, round 5) one-level dealing with a row cost per stream and quartet: sweeps on every shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5i
mkdir -p $O
export TMPDIR=/tmp
f3() { env "$@" timeout 120 python tools/form3_probe.py $ARGS 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-175; }
{
ARGS=""
echo "=== fb15k237 sum, 256 workgroups"
for sh in "1.5,1.2,0.7" "1.35,1.15,0.9"; do for rc in "0,0,0" "10,5,0" "10,10,10" "14,10,6" "16,8,0" "20,14,8"; do
    echo "shares $sh rowcost $rc: $(f3 ULTRA_STREAM_SHARES_12=$sh ULTRA_STREAM_ROW_COST_12=$rc)"; done; done
echo "=== fb15k237 sum, 192 workgroups"
for sh in "1.5,1.2,0.7" "1.35,1.15,0.9"; do for rc in "0,0,0" "10,5,0" "14,10,6"; do
    echo "shares $sh rowcost $rc: $(f3 PROBE_GRID=192 ULTRA_STREAM_SHARES_12=$sh ULTRA_STREAM_ROW_COST_12=$rc)"; done; done
ARGS="fb15k237 8 max"
echo "=== fb15k237 max"
for sh in "1.5,1.2,0.7" "1.35,1.15,0.9"; do for rc in "0,0,0" "10,5,0" "14,10,6" "28,28,28"; do
    echo "shares $sh rowcost $rc: $(f3 ULTRA_STREAM_SHARES_12=$sh ULTRA_STREAM_ROW_COST_12=$rc)"; done; done
ARGS="codex_l 8"
echo "=== codex_l sum"
for sh in "1.5,1.2,0.7" "1.35,1.15,0.9"; do for rc in "0,0,0" "10,5,0" "14,10,6" "28,28,28" "60,60,60"; do
    echo "shares $sh rowcost $rc: $(f3 ULTRA_STREAM_SHARES_12=$sh ULTRA_STREAM_ROW_COST_12=$rc)"; done; done
ARGS="codex_l 8 max"
echo "=== codex_l max"
for rc in "0,0,0" "14,10,6" "28,28,28"; do echo "rowcost $rc: $(f3 ULTRA_STREAM_ROW_COST_12=$rc)"; done
} 2>&1 | tee $O/row_cost.txt
