#!/bin/bash
# Fine-tune step: sampler one batch ahead / fused AdamW / readout gather-first / dense transposed twin, A/B on one box.
OUT=gpurun_out/r5w
mkdir -p $OUT
timeout 600 python -m pytest tests/test_training_gpu.py tests/test_models_gpu.py tests/test_rspmm_gpu.py -x -q > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 900 python - > $OUT/finetune_ab.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
for shape, combos in (("fb15k237", ((False, False), (True, False), (True, True))), ("yago310", ((False, False), (True, True)))):
    for prefetch, fused in combos:
        print(json.dumps(sb.train_case(shape, prefetch=prefetch, fused=fused)), flush=True)
print(json.dumps(sb.train_case("fb15k237", aggr="max")), flush=True)
PY
cat $OUT/finetune_ab.txt
timeout 300 python tools/train_phases.py fb15k237 > $OUT/phases.txt 2>&1; cat $OUT/phases.txt
