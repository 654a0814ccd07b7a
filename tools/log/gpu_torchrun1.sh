#!/bin/bash
# (one gpurun call) the launcher path on one GPU: bench.py under torchrun with one rank (RCCL world of 1: every step ends with its
# all-gather) beside plain bench.py; extra environment for the torchrun runs as arguments ("NCCL_MAX_NCHANNELS=2" ...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: first', round(d['ms_per_step'],4), 'repeats', d['repeats']['ms_per_step'])"; }
for cfg in "X=0" "$@"; do
env $cfg timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | pr "torchrun, 1 rank, $cfg"
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | pr "plain bench.py"
