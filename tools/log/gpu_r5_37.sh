#!/bin/bash
# The fine-tune cases inside bench.py's process (after the headline run) against a fresh process, with the sampler's stream at high priority.
O=gpurun_out/r5ai
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('in bench.py:', d['ms_per_step'], [round(c['ms_per_step'],3) for c in d['secondary']['fine_tune']])"
timeout 600 python - 2>/dev/null <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
print('fresh process:', [round(sb.train_case(s)["ms_per_step"], 3) for s in ("fb15k237", "yago310")])
PY
