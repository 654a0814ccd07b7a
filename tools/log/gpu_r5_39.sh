#!/bin/bash
# The last layer of a training step on the candidates' rows: parity tests, then the step with the route on / off.
O=gpurun_out/r5ak
mkdir -p $O
timeout 600 python -m pytest tests/test_training_gpu.py tests/test_models_gpu.py -q -x -m gpu > $O/tests.txt 2>&1
tail -12 $O/tests.txt | cut -c1-250
timeout 600 python - 2>/dev/null <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
from ultra_amd import layers
for on in (False, True):
    layers.LAST_LAYER_ON_ROWS = on
    print('last layer on rows', on, [round(sb.train_case(s)["ms_per_step"], 3) for s in ("fb15k237", "yago310")], flush=True)
PY
