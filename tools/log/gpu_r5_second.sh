#!/bin/bash
# (one gpurun call, round 5) the GPU suite with the bounded spins + measured slot streams; the form-3 probe with its per-partition
# dump; sweeps of the schedule's knobs with the new walk; the step with the slot streams chosen by measurement
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
f3() { env "$@" timeout 120 python tools/form3_probe.py 2>&1 | grep -v amdgpu.ids; }
{
echo "--- default, with the per-partition dump (whole chip, then 192 workgroups)"
f3 PROBE_DUMP_PARTS=1
f3 PROBE_DUMP_PARTS=1 PROBE_GRID=192
echo "--- quartet shares of the twelve-walker schedule"
for sh in "1.5,1.2,0.7" "1.45,1.2,0.75" "1.4,1.2,0.8" "1.4,1.15,0.85" "1.35,1.15,0.9" "1.5,1.1,0.8"; do echo "shares $sh: $(f3 ULTRA_STREAM_SHARES_12=$sh)"; done
echo "--- chain rows of up to F x the mean stream length as stream rows"
for f in 1.8 2.1 2.5 3.0 4.0; do echo "F $f: $(f3 ULTRA_CHAIN_LIMIT_FACTOR=$f)"; done
echo "--- row order inside a stream (0 by row, 1 shortest last, 2 longest last)"
for o in 0 1 2; do echo "order $o: $(f3 ULTRA_STREAM_ROW_ORDER=$o)"; done
echo "--- other shapes"
timeout 120 python tools/form3_probe.py codex_l 8 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/form3_probe.py fb15k237 8 max 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee $O/form3_sweeps.txt
timeout 300 python tools/step_probe.py 9 40 2>&1 | grep -v amdgpu.ids | tee $O/step.txt
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'], 'slot streams', d['config'].get('slot_streams'))"; }
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | pr "torchrun, 1 rank" | tee -a $O/slot_streams.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "plain bench.py" | tee -a $O/slot_streams.txt
