"""Where do the slow 20-step loops come from?  (tools/start_spread.sh: two first runs of forty sit 5 % over their process's median.)
The benchmark's timed loop -- synchronise, 20 steps through a three-deep PipelinedForward, synchronise -- repeated many times in one
process, with an event behind every step on its slot's stream and the host's time at every call: for the slowest loops, where the
time went (the GPU-side completion intervals and the host's issue times)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import secondary_bench  # noqa: E402
from ultra_amd import synthetic, tasks  # noqa: E402
from ultra_amd.graph import PipelinedForward  # noqa: E402

dev = torch.device("cuda:0")
reps, steps = int(os.environ.get("PROBE_REPS", "300")), 20
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(dev)
model = secondary_bench.load_model("sum", "ultra_3g").eval()
batches = [tasks.all_negative(data, data.target_triples[8 * i:8 * i + 8])[0] for i in range(4)]
with torch.no_grad():
    piped = PipelinedForward(model, data, batches[0], depth=3)
    piped.settle(lambda i: piped(batches[i % 4]), steps=64)
    loops = []
    for rep in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        host = []
        torch.cuda.synchronize()
        ev[0].record()
        t0 = time.perf_counter()
        for i in range(steps):
            k = piped.calls % len(piped.slots)
            piped(batches[i % 4])
            ev[i + 1].record(piped.streams[k])
            host.append(time.perf_counter() - t0)
        piped.join()
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        done = [ev[0].elapsed_time(e) for e in ev[1:]]
        loops.append((total, done, host))
tot = sorted(l[0] for l in loops)
med = tot[len(tot) // 2]
print("loops %d  ms per step: min %.4f median %.4f p90 %.4f p99 %.4f max %.4f" % (
    reps, 1e3 * tot[0] / steps, 1e3 * med / steps, 1e3 * tot[int(0.9 * reps)] / steps, 1e3 * tot[int(0.99 * reps)] / steps, 1e3 * tot[-1] / steps))
print("loops over 1.03 x median: %d of %d; first loop %.4f" % (sum(t > 1.03 * med for t in tot), reps, 1e3 * loops[0][0] / steps))
typical = min(loops, key=lambda l: abs(l[0] - med))
print("typical loop: step completion times (ms)", " ".join("%.2f" % d for d in typical[1]))
print("              host issue times (ms)     ", " ".join("%.2f" % (1e3 * h) for h in typical[2]))
for total, done, host in sorted(loops, key=lambda l: -l[0])[:4]:
    print("slow loop %.4f ms/step: completions" % (1e3 * total / steps), " ".join("%.2f" % d for d in done))
    print("                        host issue ", " ".join("%.2f" % (1e3 * h) for h in host))
