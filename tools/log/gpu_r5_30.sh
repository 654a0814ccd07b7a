#!/bin/bash
# A training step's layer as one autograd node: parity tests and step times.
OUT=gpurun_out/r5ad
mkdir -p $OUT
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_models_gpu.py tests/test_rspmm_gpu.py tests/test_layers_gpu.py -x -q > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 900 python - > $OUT/finetune.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
from ultra_amd import layers
for shape in ("fb15k237", "yago310"):
    for one in (False, True):
        layers.TRAINING_LAYER_NODE = one
        r = sb.train_case(shape)
        r["layer_as_one_node"] = one
        print(json.dumps(r), flush=True)
PY
cat $OUT/finetune.txt
