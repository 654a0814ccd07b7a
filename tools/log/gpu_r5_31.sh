#!/bin/bash
# Layer as one node, FB15k237 shape: alternating A/B in one process.
OUT=gpurun_out/r5ae
mkdir -p $OUT
timeout 900 python - > $OUT/finetune.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
from ultra_amd import layers
for one in (True, False, True, False, True, False):
    layers.TRAINING_LAYER_NODE = one
    r = sb.train_case("fb15k237")
    print(one, round(r["ms_per_step"], 3), flush=True)
PY
cat $OUT/finetune.txt
