#!/bin/bash
# (one gpurun call, round 5) A/B on one box: eight staging loads in flight (default) against four (stage4); the plain-step shortcut
# inside marker chunks (default) against without (noshortcut)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5p
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_update_gpu.py tests/test_order_gpu.py -m gpu -x -q 2>&1 | tail -2
V=$PWD/ultra_amd/lib/variants
f3() { env "$@" timeout 120 python tools/form3_probe.py $ARGS 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170; }
{
for rep in 1 2 3; do
ARGS=""
echo "default:    $(f3 X=0)"
echo "stage4:     $(f3 ULTRA_AMD_LIB=$V/libultra_amd_stage4.so)"
echo "noshortcut: $(f3 ULTRA_AMD_LIB=$V/libultra_amd_noshortcut.so)"
done
echo "--- 192 workgroups"
echo "default:    $(f3 PROBE_GRID=192)"
echo "stage4:     $(f3 PROBE_GRID=192 ULTRA_AMD_LIB=$V/libultra_amd_stage4.so)"
echo "noshortcut: $(f3 PROBE_GRID=192 ULTRA_AMD_LIB=$V/libultra_amd_noshortcut.so)"
ARGS="codex_l 8"
echo "--- codex_l"
echo "default:    $(f3 X=0)"
echo "stage4:     $(f3 ULTRA_AMD_LIB=$V/libultra_amd_stage4.so)"
echo "noshortcut: $(f3 ULTRA_AMD_LIB=$V/libultra_amd_noshortcut.so)"
} 2>&1 | tee $O/ab.txt
for v in "" stage4 noshortcut; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=$V/libultra_amd_$v.so; else unset ULTRA_AMD_LIB; fi
    echo "${v:-default}: $(timeout 300 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" | tee -a $O/ab.txt
done
