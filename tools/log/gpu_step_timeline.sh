#!/bin/bash
# per-kernel durations of the captured forward (one batch at a time), by position inside the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r4x
mkdir -p $O
export TMPDIR=/tmp
for f in ${FORMS:-0 1}; do
    rm -rf /tmp/tr$f
    (cd /tmp && ULTRA_BENCH_UPDATE_FORM=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$f -o run -- \
        python $OLDPWD/bench.py --trace-target > /dev/null 2>&1)
    find /tmp/tr$f -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_form$f.csv \;
    find /tmp/tr$f -name "*kernel_trace.csv" -exec cp {} /tmp/tr$f/trace.csv \;
    echo "== ULTRA_BENCH_UPDATE_FORM=$f"
    python - <<PY
import csv
tr = list(csv.DictReader(open("/tmp/tr$f/trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in tr]
# steps start at batch_prologue_kernel
starts = [i for i, n in enumerate(names) if "batch_prologue" in n]
last = starts[-5:]
for si in last[:2]:
    seq = tr[si:si + 60]
    t0 = int(seq[0]["Start_Timestamp"])
    for r in seq:
        if "batch_prologue" in r["Kernel_Name"] and r is not seq[0]: break
        print("%8.1f +%7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
    print("--")
PY
done
python - <<PY
import csv
tr = list(csv.DictReader(open("/tmp/tr0/trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if "rspmm_order_kernel" in r["Kernel_Name"]]
print("order kernel durations in launch order:", [round(x, 1) for x in du[:30]], "... max", max(du), "at", du.index(max(du)))
PY
