#!/bin/bash
# round 6, call 42: counters of the training step's full-graph walks at YAGO3-10's shape (three rocprofv3 --pmc passes: SQ x 2, TCC + FETCH/WRITE sizes), kernel durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_42
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  rm -rf /tmp/pmc_wk$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_wk$i -o run -- python "$OLDPWD/tools/walk_pmc_target.py" yago310 > /dev/null 2>&1)
done
python tools/summarize_pmc.py $(find /tmp/pmc_wk1 /tmp/pmc_wk2 /tmp/pmc_wk3 /tmp/pmc_wk4 /tmp/pmc_wk5 -name "*counter_collection.csv" | sort) | grep "training walk" | tee gpurun_out/r6_42/r6_pmc_training_walks.txt
python - <<'PY' | tee -a gpurun_out/r6_42/r6_pmc_training_walks.txt
import csv, glob
f = glob.glob("/tmp/pmc_wk1/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "rspmm_fwd_kernel" in n:
        d.setdefault(n[:60], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print("duration under the counter pass: %-60s %d launches, %s us" % (k, len(v), " ".join("%.0f" % t for t in v)))
PY
