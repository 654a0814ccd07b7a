"""Per-wave end of walk in the side-by-side order kernel (chain crew = waves 0..8, walkers = waves 9..15): calibrates the
schedule's stream budgets (plan.cpp COST_S_STEP_SIDE, COST_S_CHUNK).    python tools/overlap_trace.py [shape] [bs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import _lib, rspmm, synthetic  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "fb15k237"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
N, R = data.num_nodes, int(data.num_relations)
g = torch.Generator().manual_seed(0)
x = torch.randn(bs, N, 64, generator=g).to(dev)
rel = torch.randn(bs, R, 64, generator=g).to(dev)
point = (torch.arange(bs, device=dev) * 7 % N, torch.randn(bs, 64, generator=g).to(dev))
plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
grid = 256
nparts = grid // min(bs, grid)
ms, _ = plan.forward_timed(rel, x, point=point, warmup=3, iters=30)
print("%s bs %d: %.1f us per call" % (shape, bs, ms * 1e3))
trace = torch.zeros(grid * 32, dtype=torch.int64, device=dev)
_lib.check(_lib.lib.ultra_order_trace(trace.data_ptr()))
plan.forward(rel, x, point=point)
torch.cuda.synchronize()
_lib.check(_lib.lib.ultra_order_trace(None))
t = trace.cpu()
main = t[:3 * grid].view(grid, 3).double()
wave_end = t[8 * grid:24 * grid].view(grid, 16).double() - main[:, :1]
chain = main[:, 1] - main[:, 0]
chunk_ptr, _, _, _ = plan.schedule(nparts)
nchunk = (chunk_ptr[1:] - chunk_ptr[:-1]).double()
part = torch.arange(grid) // (grid // nparts)
print("cycles since the workgroup's start, mean over its %d spans; crew = waves 0..8, walkers = waves 9..15" % (grid // nparts))
print("part chunks | chain done | crew end (mean, max) | walkers end (mean, max)")
for q in range(nparts):
    sel = part == q
    we = wave_end[sel]
    print("%4d %6d | %9.0f | %9.0f %9.0f | %9.0f %9.0f" % (q, nchunk[q], chain[sel].mean(), we[:, :9].mean(), we[:, :9].max(dim=1)[0].mean(),
                                                     we[:, 9:].mean(), we[:, 9:].max(dim=1)[0].mean()))
print("all: chain %.0f, crew end %.0f, walkers end %.0f, workgroup end mean %.0f max %.0f" %
      (chain.mean(), wave_end[:, :9].mean(), wave_end[:, 9:].mean(), wave_end.max(dim=1)[0].mean(), wave_end.max()))
rel = wave_end - chain.unsqueeze(1)
print("end of walk per wave, cycles since the workgroup's chains were done (mean over workgroups):")
print(" ".join("%6.0f" % v for v in rel.mean(dim=0).tolist()))
sel = nchunk[part] > 4
if sel.any():
    A = torch.stack([nchunk[part][sel], torch.ones(int(sel.sum()), dtype=torch.double)], dim=1)
    sol = torch.linalg.lstsq(A, chain[sel].unsqueeze(1)).solution.flatten()
    print("chain fit: %.0f cycles per chunk + %.0f" % (sol[0], sol[1]))
