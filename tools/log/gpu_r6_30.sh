#!/bin/bash
# round 6, call 30: prefetch_negatives waits for its first observations: the fine-tune cases of tools/secondary_bench.py (standalone, three processes) and of bench.py,
# each with what the sampler's stream check saw
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_30
cat > /tmp/summ.py <<'PY'
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if 'case' in d and 'fine-tune' in d['case']:
            print('secondary_bench', d['shape'], d['aggregate'], round(d['ms_per_step'], 3), round(d['ms_per_step_eager'], 3), d.get('sampler_stream_check'))
        if 'secondary' in d:
            print('bench ms_per_step', d['ms_per_step'])
            for c in d['secondary']['fine_tune']:
                print('bench', c['shape'], c['aggregate'], round(c['ms_per_step'], 3), round(c['ms_per_step_eager'], 3), c.get('sampler_stream_check'))
PY
for i in 1 2 3; do timeout 600 python tools/secondary_bench.py 2>/dev/null | python /tmp/summ.py; done | tee gpurun_out/r6_30/out.txt
timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python /tmp/summ.py | tee -a gpurun_out/r6_30/out.txt
