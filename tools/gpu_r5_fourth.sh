#!/bin/bash
# (one gpurun call, round 5) where the form-3 layer's time goes, timing-only variant libraries (wrong results on purpose):
# nomfma (a VALU add per operand instead of the matrix instructions), noupd (no operand reads, no matrix chains), mfma32 (the same
# matrix-pipe time as half as many 32x32x2 instructions); + how often the walkers find the hand-off ring full
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5d
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_update_gpu.py tests/test_order_gpu.py -m gpu -x -q 2>&1 | tail -3
{
for rep in 1 2; do
for v in "" mfma32 nomfma noupd; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=$PWD/ultra_amd/lib/variants/libultra_amd_$v.so; else unset ULTRA_AMD_LIB; fi
    echo "${v:-default}: $(timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1)"
done
done
unset ULTRA_AMD_LIB
echo "--- 192 workgroups"
for v in "" mfma32 nomfma; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=$PWD/ultra_amd/lib/variants/libultra_amd_$v.so; else unset ULTRA_AMD_LIB; fi
    echo "${v:-default}: $(PROBE_GRID=192 timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1)"
done
unset ULTRA_AMD_LIB
echo "--- codex_l, max"
timeout 120 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids | tail -1
timeout 120 python tools/form3_probe.py fb15k237 8 max 2>&1 | grep -v amdgpu.ids | tail -1
} 2>&1 | tee $O/form3_variants.txt
