#!/bin/bash
# does the kernel tracer see the durations HIP events see?  tools/beside_probe.py under rocprofv3 --kernel-trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r4y
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/try
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/try -o run -- \
    python $OLDPWD/tools/beside_probe.py fb15k237 8 > $O/probe_traced.txt 2>&1)
grep -E "hipGraph|back to back" $O/probe_traced.txt
find /tmp/try -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_probe.csv \;
find /tmp/try -name "*kernel_trace.csv" -exec cp {} /tmp/try/trace.csv \;
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/kernel_stats_probe.csv")))
for r in rows[:8]:
    print("%-100s calls %5d avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:100], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
# distribution of form-3 durations in launch order
tr = list(csv.DictReader(open("/tmp/try/trace.csv")))
d = collections.defaultdict(list)
for r in tr:
    d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in d.items():
    if "rspmm_order_kernel" in k:
        v.sort()
        du = [(e - s) / 1e3 for s, e in v]
        gaps = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
        print(k[-40:], "n", len(du), "first 10:", [round(x, 1) for x in du[:10]], "median %.1f" % sorted(du)[len(du) // 2],
              "gap median %.1f" % sorted(gaps)[len(gaps) // 2])
PY
