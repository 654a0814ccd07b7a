#!/bin/bash
# Collect the round's measurement artefacts on a GPU box into gpurun_out/<round>/ (copied to profiles/ afterwards).
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh A r2'   smoke, the bench line (+ raw PMC passes), kernel stats of the
#                                                                    judged command, 1-process torchrun line
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh B r2'   secondary cases, fine-tune kernel stats, kernel probes,
#                                                                    evaluation protocol speed
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
part="${1:-A}"
R="${2:-r2}"
OUT=gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$part" = "A" ]; then
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/${R}_smoke.txt" 2>&1
    ULTRA_BENCH_PMC_KEEP="$PWD/$OUT/${R}_pmc" timeout 900 python bench.py > "$OUT/${R}_bench.json" 2> "$OUT/bench.err"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o run -- \
        python "$OLDPWD/bench.py" --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1)
    find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/${R}_bench_kernel_stats.csv" \;
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/${R}_bench_torchrun1.json" 2>> "$OUT/bench.err"
else
    timeout 600 python tools/secondary_bench.py > "$OUT/${R}_secondary.jsonl" 2> "$OUT/secondary.err"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o run -- \
        python "$OLDPWD/tools/train_probe.py" > /dev/null 2>&1)
    find /tmp/prof_train -name "*kernel_stats.csv" -exec cp {} "$OUT/${R}_finetune_kernel_stats.csv" \;
    { timeout 120 python tools/conv_probe.py; timeout 120 python tools/readout_probe.py; } > "$OUT/${R}_kernel_probes.txt" 2>&1
    timeout 600 python tools/eval_speed.py 512 > "$OUT/${R}_eval_speed.txt" 2>&1
fi
ls -la "$OUT"
