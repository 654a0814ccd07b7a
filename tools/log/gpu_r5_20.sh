#!/bin/bash
# (one gpurun call, round 5) does the walk want more gathers in flight?  Proxy: FEWER walkers with all the rows (per-wave shares: a share of 0
# leaves a wave without rows) -- if eight or ten walkers walk as fast as twelve, a deeper queue per lane would not help either.
# Then collections A and C of the final tree (the second run of this script did only those: the first one's redirect met a missing directory).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5t
mkdir -p $O gpurun_out/r5
export TMPDIR=/tmp
f3() { env "$@" timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-175; }
{
echo "twelve walkers (default):            $(f3 X=0)"
echo "ten walkers (waves 10, 11 idle):      $(f3 ULTRA_STREAM_SHARES_WAVES_12=1.4,1.4,1.4,1.4,1.2,1.2,1.2,1.2,1.0,1.0,0,0)"
echo "eight walkers (quartet 2 idle):       $(f3 ULTRA_STREAM_SHARES_WAVES_12=1.6,1.6,1.6,1.6,1.3,1.3,1.3,1.3,0,0,0,0)"
echo "eight walkers, flatter shares:        $(f3 ULTRA_STREAM_SHARES_WAVES_12=1.5,1.5,1.5,1.5,1.4,1.4,1.4,1.4,0,0,0,0)"
echo "six walkers (waves 0, 1 of each quartet): $(f3 ULTRA_STREAM_SHARES_WAVES_12=1.5,1.5,0,0,1.3,1.3,0,0,1.0,1.0,0,0)"
echo "twelve walkers again:                 $(f3 X=0)"
} 2>&1 | tee $O/fewer_walkers.txt
bash tools/collect_profiles.sh A r5 > gpurun_out/r5/collect_A.log 2>&1
bash tools/collect_profiles.sh C r5 > gpurun_out/r5/collect_C.log 2>&1
tail -2 gpurun_out/r5/collect_C.log
