#!/bin/bash
# round 6, call 28: the sampler's stream re-checked while the loop runs: secondary.fine_tune inside bench.py's process; the probe's variants; tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_28
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_train_gpu.py tests/test_launch_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r6_28/bench.json 2> gpurun_out/r6_28/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_28/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"])
for c in d["secondary"]["fine_tune"]:
    print(c["shape"], c["aggregate"], round(c["ms_per_step"], 3), round(c["ms_per_step_eager"], 3))
PY
for v in a c d; do PROBE_VARIANT=$v timeout 300 python tools/train_after_pipeline_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; done
