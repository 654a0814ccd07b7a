#!/bin/bash
# (one gpurun call, round 5, final tree) tools/collect_profiles.sh A: smoke, the bench line with its secondary block and PMC passes, kernel
# stats of the judged command, one launcher's rank.  (B's files come from tools/gpu_r5_final2.sh, the suite log from tools/gpu_r5_36.sh.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5
mkdir -p $O
export TMPDIR=/tmp
bash tools/collect_profiles.sh A r5 > $O/collect_A.log 2>&1
python -c "
import json
d=json.loads(open('$O/r5_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), [round(c['ms_per_step'],3) for c in d['secondary']['fine_tune']], d['roofline']['achieved'], d['roofline']['frac'])
t=json.loads(open('$O/r5_bench_torchrun1.json').read().strip().splitlines()[-1])
print('torchrun1', t['ms_per_step'], t.get('ms_per_step_median'))"
cat $O/r5_smoke.txt | tail -2
