#!/bin/bash
# (one gpurun call, round 5, after the fine-tuning work) GPU suite on the final tree, tools/collect_profiles.sh A (bench line with its
# secondary block, kernel stats, one launcher's rank) and B (secondary cases, fine-tune / config-3 kernel stats, probes, evaluation speed).
# The counter passes (C), form-3 probes and launcher-path lines of tools/gpu_r5_final.sh were not repeated: the inference path is unchanged.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/r5_gpu_tests.txt 2>&1
tail -3 $O/r5_gpu_tests.txt
bash tools/collect_profiles.sh A r5 > $O/collect_A.log 2>&1
bash tools/collect_profiles.sh B r5 > $O/collect_B.log 2>&1
timeout 300 python tools/train_phases.py fb15k237 > $O/r5_finetune_phases.txt 2>&1
timeout 300 python tools/train_phases.py yago310 >> $O/r5_finetune_phases.txt 2>&1
python -c "
import json
d=json.loads(open('$O/r5_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['secondary']['fine_tune'] if 'secondary' in d else None)"
cat $O/r5_secondary.jsonl | cut -c1-300
ls $O
