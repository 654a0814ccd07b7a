#!/bin/bash
# Strict sampler as a kernel: parity with the mask formulation, then the fine-tune step in a fresh process and inside bench.py.
O=gpurun_out/r5aj
mkdir -p $O
timeout 600 python -m pytest tests/test_training_gpu.py tests/test_models_gpu.py -q -x -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt | cut -c1-200
timeout 600 python - 2>/dev/null <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
from ultra_amd import tasks
for kernel in (False, True):
    tasks.STRICT_SAMPLER_KERNEL = kernel
    print('sampler kernel', kernel, [round(sb.train_case(s)["ms_per_step"], 3) for s in ("fb15k237", "yago310")], flush=True)
PY
