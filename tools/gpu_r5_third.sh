#!/bin/bash
# (one gpurun call, round 5) GPU suite after the removals (form 2, side-by-side chain build); per-wave dumps of form 3 for the
# schedule's cost model (tools/wave_fit.py); first sweeps of the per-quartet row cost
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5c
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
f3() { env "$@" timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids; }
{
echo "--- dumps"
f3 PROBE_DUMP_PARTS=1
f3 PROBE_DUMP_PARTS=1 PROBE_GRID=192
PROBE_DUMP_PARTS=1 timeout 120 python tools/form3_probe.py fb15k237 8 max 2>&1 | grep -v amdgpu.ids
PROBE_DUMP_PARTS=1 timeout 120 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids
} > $O/form3_dumps.txt 2>&1
grep -v "^PARTS" $O/form3_dumps.txt
{
echo "--- row cost per quartet (steps a row), default shares 1.5,1.2,0.7"
for rc in "0,0,0" "10,5,0" "20,8,0" "30,10,0" "45,12,0" "56,15,0" "20,10,5" "12,12,12"; do echo "rowcost $rc: $(f3 ULTRA_STREAM_ROW_COST_12=$rc)"; done
echo "--- ... with shares 1.35,1.15,0.9"
for rc in "0,0,0" "10,5,0" "20,8,0" "30,10,0"; do echo "rowcost $rc: $(f3 ULTRA_STREAM_ROW_COST_12=$rc ULTRA_STREAM_SHARES_12=1.35,1.15,0.9)"; done
echo "--- shares, finer"
for sh in "1.3,1.15,0.95" "1.3,1.1,1.0" "1.25,1.15,1.0" "1.2,1.2,1.0"; do echo "shares $sh: $(f3 ULTRA_STREAM_SHARES_12=$sh)"; done
} 2>&1 | tee $O/form3_rowcost.txt
