"""Two batches in flight: ms per step of `runs` consecutive runs of 20 steps (5 warm-up steps and a synchronisation before each, as
bench.py does) by the number of workgroups the aggregation kernels of each capture are launched with.
    python tools/share_probe.py [runs] [grid ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import models, synthetic, tasks
from ultra_amd.graph import GraphedForward, PipelinedForward

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
grids = [int(g) for g in sys.argv[2:]] or [256, 192, 128]
dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(dev)
model = models.Ultra(**synthetic.default_model_cfg())
model.load_state_dict(torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ultra_3g_model.pt")))
model = model.to(dev).eval()
triples = data.target_triples
inputs = [tasks.all_negative(data, triples[8 * i:8 * i + 8])[0] for i in range(16)]
for grid in grids:
    pf = PipelinedForward(model, data, inputs[0], depth=2,
                          slot_factory=lambda: GraphedForward(model, data, inputs[0], launch_grid=0 if grid == 256 else grid))
    out = []
    with torch.no_grad():
        for r in range(runs):
            for i in range(5):
                pf(inputs[i % 16])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20):
                pf(inputs[(5 + i) % 16])
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 20 * 1e3)
    print("grid %3d: " % grid + " ".join("%.4f" % v for v in out) + "   median %.4f max %.4f" % (sorted(out)[len(out) // 2], max(out)))
    del pf
