#!/bin/bash
# First-run figures of N fresh bench.py processes, plain and as one rank under torch.distributed.run:
#   bash tools/start_spread.sh [N] > profiles/rX_start_spread.txt
# Each start: bench.py --steps 20 --warmup 5 without the roofline / secondary / cpu-baseline blocks (the timed region and what runs in front
# of it -- captures, slot-stream trial, 64 settle steps, 5 warm-up steps -- are the driver command's).
N=${1:-20}
cd $(dirname $0)/..
summ='
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
s = (d["config"].get("slot_streams") or {})
print("%.4f  median %.4f  repeats %s  chosen %s #%s  settle %s" % (d["ms_per_step"], d["ms_per_step_median"], d["repeats"]["ms_per_step"], s.get("chosen"), s.get("candidate"), s.get("settle")))
'
for mode in plain torchrun; do
  echo "== $mode: $N fresh processes, ms per step (first run of 20 steps)"
  for i in $(seq 1 $N); do
    if [ $mode = plain ]; then
      timeout 600 python bench.py --steps 20 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline 2>/dev/null | python -c "$summ"
    else
      port=$((20000 + RANDOM % 20000))
      timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 20 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline 2>/dev/null | python -c "$summ"
    fi
  done
done
