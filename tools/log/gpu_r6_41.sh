#!/bin/bash
# round 6, call 41: SQ counters of the update backward at 985 k rows (two passes, counters only + kernel trace), summarised by tools/summarize_pmc.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6_41
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY"; do
  i=$((i + 1))
  rm -rf /tmp/pmc_cb$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_cb$i -o run -- python "$OLDPWD/tools/conv_bwd_probe.py" 985456 > /dev/null 2>&1)
done
python tools/summarize_pmc.py $(find /tmp/pmc_cb1 /tmp/pmc_cb2 -name "*counter_collection.csv" | sort) | tee gpurun_out/r6_41/r6_pmc_update_backward.txt
