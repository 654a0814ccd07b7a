#!/bin/bash
# (one gpurun call, round 5) the workgroup-local pool of the twelve-walker schedule: fraction of every stream's steps, row lengths
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5o
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_update_gpu.py -m gpu -x -q 2>&1 | tail -2
ULTRA_POOL_12=0.12,12,24 timeout 600 python -m pytest tests/test_fused_update_gpu.py tests/test_baseline_parity_gpu.py -m gpu -x -q 2>&1 | tail -2
f3() { env "$@" timeout 120 python tools/form3_probe.py $ARGS 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170; }
{
ARGS=""
echo "no pool: $(f3 X=0)"
for w in "8,16" "12,24" "16,32" "24,48"; do for f in 0.06 0.10 0.15 0.22; do
    echo "pool $f of the steps, rows of $w edges: $(f3 ULTRA_POOL_12=$f,$w)"; done; done
echo "--- with the quartet shares of round 4 (1.5,1.2,0.7) and equal-ish ones"
for sh in "1.5,1.2,0.7" "1.2,1.1,1.0"; do for f in 0.10 0.15; do
    echo "shares $sh pool $f,12,24: $(f3 ULTRA_POOL_12=$f,12,24 ULTRA_STREAM_SHARES_12=$sh)"; done; done
echo "--- 192 workgroups"
echo "no pool: $(f3 PROBE_GRID=192)"
for f in 0.06 0.10 0.15; do echo "pool $f,12,24: $(f3 PROBE_GRID=192 ULTRA_POOL_12=$f,12,24)"; done
ARGS="codex_l 8"
echo "--- codex_l"
echo "no pool: $(f3 X=0)"
for w in "8,16" "12,24"; do for f in 0.10 0.15; do echo "pool $f,$w: $(f3 ULTRA_POOL_12=$f,$w)"; done; done
ARGS="fb15k237 8 max"
echo "--- fb15k237 max"
echo "no pool: $(f3 X=0)"
echo "pool 0.10,12,24: $(f3 ULTRA_POOL_12=0.10,12,24)"
} 2>&1 | tee $O/pool.txt
