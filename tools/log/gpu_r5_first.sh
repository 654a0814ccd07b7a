#!/bin/bash
# (one gpurun call, round 5) the GPU suite on the new walk; then A/B on one box: the walk's instruction diet against round 4's loops
# (variant library r4walk: tools/build_variant.py with the generator's ULTRA_GEN_DIET=0 header), and the pipeline slots' stream
# priority on the plain and the launcher's path.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
tail -5 $O/gpu_tests.txt
for v in "" r4walk; do
    if [ -n "$v" ]; then export ULTRA_AMD_LIB=$PWD/ultra_amd/lib/variants/libultra_amd_$v.so; else unset ULTRA_AMD_LIB; fi
    echo "=== library: ${v:-default (diet)}"
    timeout 300 python tools/beside_probe.py fb15k237 8 2>&1 | grep -v amdgpu.ids > $O/beside_${v:-diet}.txt
    grep -E "two launches|hipGraph|^form 3|all equal" $O/beside_${v:-diet}.txt
    timeout 300 python tools/step_probe.py 9 40 2>&1 | grep -v amdgpu.ids | tee $O/step_${v:-diet}.txt
done
unset ULTRA_AMD_LIB
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'])"; }
for cfg in "ULTRA_SLOT_STREAM_PRIORITY=-1" "ULTRA_SLOT_STREAM_PRIORITY=0" "ULTRA_SLOT_STREAM_PRIORITY=0 ULTRA_BENCH_LAUNCHER_QUEUES=3"; do
env $cfg timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | pr "torchrun, 1 rank, $cfg" | tee -a $O/slot_streams.txt
done
for cfg in "ULTRA_SLOT_STREAM_PRIORITY=-1" "ULTRA_SLOT_STREAM_PRIORITY=0"; do
env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "plain bench.py, $cfg" | tee -a $O/slot_streams.txt
done
