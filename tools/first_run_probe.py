"""Why is the first timed run of bench.py ~4 % slower than the repeats?  Same loop as bench.py (two captured forwards in flight),
variants: fresh inputs per step vs one input; extra graph replays before the first run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import models, synthetic, tasks
from ultra_amd.graph import PipelinedForward

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(dev)
model = models.Ultra(**synthetic.default_model_cfg())
model.load_state_dict(torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ultra_3g_model.pt")))
model = model.to(dev).eval()
triples = data.target_triples
mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
n_in = 1 if mode == "one_input" else 25
inputs = [tasks.all_negative(data, triples[8 * i:8 * i + 8])[0] for i in range(n_in)]
pf = PipelinedForward(model, data, inputs[0], depth=2)
if mode == "prereplay":
    for s in pf.slots:
        for _ in range(4):
            s.graph.replay()
    torch.cuda.synchronize()


def run(W, K, base):
    with torch.no_grad():
        for i in range(W):
            pf(inputs[(base + i) % n_in])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            pf(inputs[(base + W + i) % n_in])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3


print(mode, " ".join("%.4f" % run(5, 20, 0) for _ in range(6)))
# per-step device time of a run right after: events on the slot streams
