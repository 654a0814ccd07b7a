#!/bin/bash
# (one gpurun call, round 5, FINAL tree: after the strict-sampler kernel) the GPU suite, tools/collect_profiles.sh A and B, the fine-tune phases.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/r5_gpu_tests.txt 2> $O/r5_gpu_tests.err
tail -2 $O/r5_gpu_tests.txt | cut -c1-200
bash tools/collect_profiles.sh A r5 > $O/collect_A.log 2>&1
bash tools/collect_profiles.sh B r5 > $O/collect_B.log 2>&1
timeout 300 python tools/train_phases.py fb15k237 > $O/r5_finetune_phases.txt 2>&1
timeout 300 python tools/train_phases.py yago310 >> $O/r5_finetune_phases.txt 2>&1
python -c "
import json
d=json.loads(open('$O/r5_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), [round(c['ms_per_step'],3) for c in d['secondary']['fine_tune']], d['roofline']['achieved'], d['roofline']['frac'])
t=json.loads(open('$O/r5_bench_torchrun1.json').read().strip().splitlines()[-1])
print('torchrun1', t['ms_per_step'], t.get('ms_per_step_median'))"
grep -h "fine-tune" $O/r5_secondary.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['aggregate'], round(d['ms_per_step'],3))"
grep -v amdgpu $O/r5_finetune_phases.txt | cut -c1-400
