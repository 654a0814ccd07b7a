#!/bin/bash
# (one gpurun call, round 5) where the form-3 layer's time goes, unperturbed: timing-only variant libraries (wrong results on
# purpose) nomfma / noupd / mfma32, the park-retry count (parkdiag: one LDS count per retry and wave), and the relation-graph layer
# with one and two row tiles per workgroup (whole chip / beside an entity layer)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5e
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_order_gpu.py -m gpu -x -q -k "dense" 2>&1 | tail -3
{
for rep in 1 2; do
for v in "" parkdiag mfma32 nomfma noupd; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=$PWD/ultra_amd/lib/variants/libultra_amd_$v.so; else unset ULTRA_AMD_LIB; fi
    echo "${v:-default}: $(timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1)"
done
done
echo "--- 192 workgroups"
for v in "" parkdiag mfma32 nomfma; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=$PWD/ultra_amd/lib/variants/libultra_amd_$v.so; else unset ULTRA_AMD_LIB; fi
    echo "${v:-default}: $(PROBE_GRID=192 timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1)"
done
echo "--- codex_l, max (parkdiag)"
export ULTRA_AMD_LIB=$PWD/ultra_amd/lib/variants/libultra_amd_parkdiag.so
timeout 120 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py fb15k237 8 max 2>&1 | grep -v amdgpu.ids | tail -1
unset ULTRA_AMD_LIB
echo "--- relation-graph layer: one tile per workgroup (240 workgroups) / two (120)"
timeout 120 python tools/dense_order_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
PROBE_GRID=192 timeout 120 python tools/dense_order_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
} 2>&1 | tee $O/form3_variants.txt
timeout 300 python tools/step_probe.py 9 40 2>&1 | grep -v amdgpu.ids | tee $O/step.txt
ULTRA_DOL_TILES=1 timeout 300 python tools/step_probe.py 9 40 2>&1 | grep -v amdgpu.ids | sed 's/^/one tile per relation-layer workgroup: /' | tee -a $O/step.txt
