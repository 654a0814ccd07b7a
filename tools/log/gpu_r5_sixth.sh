#!/bin/bash
# (one gpurun call, round 5) where the update waves sit: one per SIMD (default) / two on SIMDs 2 and 3 (upd2) / all on SIMD 3 (upd1),
# with per-wave shares of the stream work
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5f
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_order_gpu.py -m gpu -x -q -k "dense" 2>&1 | tail -2
f3() { env "$@" timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; }
V=$PWD/ultra_amd/lib/variants
{
echo "default: $(f3 X=0)"
echo "--- upd2: update waves on SIMDs 2, 3 (physical 10, 11, 14, 15); walkers 2, 3, 6, 7 share those SIMDs"
echo "quartet shares: $(f3 ULTRA_AMD_LIB=$V/libultra_amd_upd2.so)"
for sh in "1.6,1.6,1.0,1.0,1.4,1.4,0.8,0.8,0.9,0.9,0.9,0.9" "1.5,1.5,0.8,0.8,1.3,1.3,0.6,0.6,1.0,1.0,1.0,1.0" "1.7,1.7,1.2,1.2,1.4,1.4,0.9,0.9,0.7,0.7,0.7,0.7" \
          "1.6,1.6,0.6,0.6,1.4,1.4,0.5,0.5,1.1,1.1,1.1,1.1" "1.8,1.8,1.0,1.0,1.5,1.5,0.7,0.7,0.8,0.8,0.8,0.8"; do
    echo "waves $sh: $(f3 ULTRA_AMD_LIB=$V/libultra_amd_upd2.so ULTRA_STREAM_SHARES_WAVES_12=$sh)"
done
echo "--- upd1: all update waves on SIMD 3; walkers (logical 3 k + j = physical 4 k + j) on SIMDs 0 .. 2"
echo "quartet shares: $(f3 ULTRA_AMD_LIB=$V/libultra_amd_upd1.so)"
for sh in "1.6,1.6,1.6,1.2,1.2,1.2,0.8,0.8,0.8,0.4,0.4,0.4" "1.5,1.5,1.5,1.2,1.2,1.2,0.9,0.9,0.9,0.5,0.5,0.5" "1.3,1.3,1.3,1.1,1.1,1.1,0.9,0.9,0.9,0.7,0.7,0.7" \
          "1.8,1.8,1.8,1.2,1.2,1.2,0.7,0.7,0.7,0.3,0.3,0.3"; do
    echo "waves $sh: $(f3 ULTRA_AMD_LIB=$V/libultra_amd_upd1.so ULTRA_STREAM_SHARES_WAVES_12=$sh)"
done
echo "default again: $(f3 X=0)"
} 2>&1 | tee $O/update_simds.txt
