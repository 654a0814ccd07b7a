#!/bin/bash
# (one gpurun call, round 5) the slots' streams picked among eight candidate sets: plain process and one launcher's rank, two / three in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5s
mkdir -p $O
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'], 'slot streams', d['config'].get('slot_streams'))"; }
{
for rep in 1 2; do for k in 3 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --in-flight $k --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "plain, in flight $k"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 20 --warmup 5 --in-flight $k --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "torchrun, 1 rank, in flight $k"
done; done
echo "the driver's command: $(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | pr 'python bench.py --gpus 1 --steps 20 --warmup 5')"
for d in 3 2; do echo "step_probe depth $d: $(PROBE_DEPTH=$d timeout 300 python tools/step_probe.py 7 40 2>&1 | grep -v amdgpu.ids | tail -1)"; done
} 2>&1 | tee $O/stream_search.txt
timeout 600 python -m pytest tests/test_models_gpu.py tests/test_eval_gpu.py tests/test_launch_gpu.py -m gpu -x -q 2>&1 | tail -3
