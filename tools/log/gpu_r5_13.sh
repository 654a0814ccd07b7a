#!/bin/bash
# (one gpurun call, round 5) batches in flight: 2 / 3 / 4 in tools/step_probe.py and in bench.py (plain and one launcher's rank)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5m
mkdir -p $O
export TMPDIR=/tmp
{
for d in 2 3 4 3 2; do echo "depth $d: $(PROBE_DEPTH=$d timeout 300 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu.ids | tail -1)"; done
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'], 'slot streams', d['config'].get('slot_streams'))"; }
for k in 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --in-flight $k --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "plain bench.py --in-flight $k"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 20 --warmup 5 --in-flight $k --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "torchrun, 1 rank, --in-flight $k"
done
} 2>&1 | tee $O/in_flight.txt
