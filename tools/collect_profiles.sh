#!/bin/bash
# Collect the round's measurement artefacts on a GPU box into gpurun_out/<round>/ (copied to profiles/ afterwards).
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh A r2'   smoke, the bench line (+ raw PMC passes), kernel stats of the
#                                                                    judged command, 1-process torchrun line
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh B r2'   secondary cases, fine-tune and config-3 kernel stats, kernel probes,
#                                                                    evaluation protocol speed
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh C r2'   SQ / TCC counter passes over tools/pmc_target.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
part="${1:-A}"
R="${2:-r6}"
OUT=gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$part" = "A" ]; then
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/${R}_smoke.txt" 2>&1
    ULTRA_BENCH_PMC_KEEP="$PWD/$OUT/${R}_pmc" timeout 900 python bench.py > "$OUT/${R}_bench.json" 2> "$OUT/bench.err"
    # (bench.py's own --kernel-trace child pass: the forward as ONE captured graph on ONE stream -- the durations DESIGN.md section 4 quotes)
    # (bench.py's own --kernel-trace child passes: inflight1 = the forward as ONE captured graph on ONE stream -- the durations DESIGN.md
    # section 4 quotes; inflight<k> = k captures on k streams as the timed region runs them)
    for f in "$OUT/${R}_pmc"/bench_kernel_stats_inflight*.csv; do cp "$f" "$OUT/${R}_$(basename "$f")" 2>/dev/null; done
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o run -- \
        python "$OLDPWD/bench.py" --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1)
    find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/${R}_bench_kernel_stats.csv" \;
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/${R}_bench_torchrun1.json" 2>> "$OUT/bench.err"
elif [ "$part" = "C" ]; then
    # SQ / TCC counter passes over tools/pmc_target.py (separate passes, no other trace domain)
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
               "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
        i=$((i + 1))
        (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_sq$i -o run -- \
            python "$OLDPWD/tools/pmc_target.py" > /dev/null 2>&1)
    done
    python tools/summarize_pmc.py $(find /tmp/pmc_sq1 /tmp/pmc_sq2 /tmp/pmc_sq3 -name "*counter_collection.csv" | sort) > "$OUT/${R}_pmc_sq.txt" 2>&1
else
    timeout 600 python tools/secondary_bench.py > "$OUT/${R}_secondary.jsonl" 2> "$OUT/secondary.err"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o run -- \
        python "$OLDPWD/tools/train_probe.py" > /dev/null 2>&1)
    find /tmp/prof_train -name "*kernel_stats.csv" -exec cp {} "$OUT/${R}_finetune_kernel_stats.csv" \;
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o run -- \
        python "$OLDPWD/tools/config3_probe.py" max > /dev/null 2>&1)
    find /tmp/prof_c3 -name "*kernel_stats.csv" -exec cp {} "$OUT/${R}_config3_kernel_stats.csv" \;
    { timeout 120 python tools/conv_probe.py; timeout 120 python tools/readout_probe.py; } > "$OUT/${R}_kernel_probes.txt" 2>&1
    timeout 600 python tools/eval_speed.py 512 > "$OUT/${R}_eval_speed.txt" 2>&1
    # the fine-tuning step launched kernel by kernel and as one hipGraph replay: ms per step and host issue time of both forms
    timeout 600 python tools/train_graph_probe.py fb15k237 yago310 2>&1 | grep -v amdgpu.ids > "$OUT/${R}_finetune_phases.txt"
    for shape in fb15k237 yago310; do
        rm -rf /tmp/tl_$shape
        (cd /tmp && PROBE_ONLY=eager timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$shape -- \
            python "$OLDPWD/tools/train_graph_probe.py" $shape > /dev/null 2>&1)
        python tools/train_timeline.py $(ls /tmp/tl_$shape/*/*_kernel_trace.csv | head -1) > "$OUT/${R}_finetune_timeline_$shape.txt" 2>&1
    done
fi
ls -la "$OUT"
