#!/bin/bash
# round 6: 20 fresh starts plain + 20 under torch.distributed.run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/start_spread.sh 20 > gpurun_out/r6_start_spread.txt 2>&1
python - <<'PY'
import re
for mode, block in re.findall(r"== (\w+):.*?\n((?:[0-9.]+ .*\n?)+)", open("gpurun_out/r6_start_spread.txt").read()):
    v = sorted(float(l.split()[0]) for l in block.strip().splitlines())
    med = v[len(v) // 2]
    print(mode, "n", len(v), "min %.4f median %.4f max %.4f  (max - min) / median = %.2f %%" % (v[0], med, v[-1], 100 * (v[-1] - v[0]) / med))
PY
