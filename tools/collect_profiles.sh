#!/bin/bash
# Collect the round's measurement artefacts on a GPU box into gpurun_out/ (copied to profiles/ afterwards).
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh A'      bench lines, kernel stats, smoke, 1-process torchrun
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh B'      roofline points, HBM-traffic PMC passes, parity, eval, secondary
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh C'      SQ / TCC counter passes over tools/pmc_target.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r1
mkdir -p "$OUT"
export TMPDIR=/tmp
part="${1:-A}"
if [ "$part" = "A" ]; then
    python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.txt" 2>&1
    python bench.py --steps 50 --warmup 5 > "$OUT/r1_bench.json" 2> "$OUT/bench.err"
    python bench.py --steps 50 --warmup 5 --no-graph --no-cpu-baseline > "$OUT/r1_bench_eager.json" 2>> "$OUT/bench.err"
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o run -- \
        python "$OLDPWD/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
    find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/r1_bench_kernel_stats.csv" \;
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/r1_bench_torchrun1.json" 2>> "$OUT/bench.err"
elif [ "$part" = "C" ]; then
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
               "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
        i=$((i + 1))
        (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_sq$i -o run -- \
            python "$OLDPWD/tools/pmc_target.py" > /dev/null 2>&1)
    done
    python tools/summarize_pmc.py $(find /tmp/pmc_sq1 /tmp/pmc_sq2 /tmp/pmc_sq3 -name "*counter_collection.csv" | sort) > "$OUT/r1_pmc_sq.txt" 2>&1
else
    python tools/roofline_points.py --out "$OUT/r1_roofline_points.json" > "$OUT/roofline_points.txt" 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o run -- \
            python "$OLDPWD/tools/roofline_points.py" --pmc --out /tmp/rp_$c.json > /dev/null 2>&1)
        mkdir -p "$OUT/r1_pmc"
        find /tmp/pmc_$c -name "*counter_collection.csv" -exec cp {} "$OUT/r1_pmc/${c}_counter_collection.csv" \;
    done
    python tools/parse_pmc.py "$OUT/r1_pmc/FETCH_SIZE_counter_collection.csv" "$OUT/r1_pmc/WRITE_SIZE_counter_collection.csv" \
        "$OUT/r1_rspmm_hbm_traffic.json" > "$OUT/parse_pmc.txt" 2>&1
    python tools/parity_report.py > "$OUT/parity.txt" 2>&1; cp gpurun_out/parity_report.json "$OUT/r1_parity_report.json"
    python tools/eval_speed.py 512 > "$OUT/r1_eval_speed.txt" 2>&1
    python tools/secondary_bench.py > "$OUT/r1_secondary.jsonl" 2> "$OUT/secondary.err"
fi
ls -la "$OUT"
