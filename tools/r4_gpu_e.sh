#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4e
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -5 $O/tests_all.txt
timeout 600 python - > $O/config3.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
import secondary_bench as sb
for aggr in ("max", "sum"):
    print(json.dumps(sb.forward_parity_case("codex_l", aggr, "ultra_50g", n_batch=1)), flush=True)
print(json.dumps(sb.forward_parity_case("fb15k237", "max", "ultra_50g", n_batch=1)), flush=True)
PY
cut -c1-330 $O/config3.txt
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o run -- \
    python "$OLDPWD/tools/config3_probe.py" max > /dev/null 2>&1)
find /tmp/prof_c3 -name "*kernel_stats.csv" -exec cp {} "$O/config3_kernel_stats.csv" \;
head -12 $O/config3_kernel_stats.csv | cut -c1-150
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o run -- \
    python "$OLDPWD/tools/train_probe.py" > /dev/null 2>&1)
find /tmp/prof_train -name "*kernel_stats.csv" -exec cp {} "$O/finetune_kernel_stats.csv" \;
head -40 $O/finetune_kernel_stats.csv | cut -c1-150
