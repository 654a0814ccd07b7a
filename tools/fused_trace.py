"""Shader-clock timeline of the one-launch layer (ultra_rspmm_forward_update) per workgroup: walk phases and the update tail.
    python tools/fused_trace.py [shape] [bs]        (clock = s_memtime ticks, 100 MHz)"""
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
w = (torch.randn(64, 128, generator=g) / 11).to(dev)
b, lw, lb = (torch.randn(64, generator=g).to(dev) for _ in range(3))
plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
grid = 256
for _ in range(3):
    plan.forward_update(rel, x, w, b, lw, lb, 1e-5, 7, point=point)
torch.cuda.synchronize()
trace = torch.zeros(grid * 32, dtype=torch.int64, device=dev)
_lib.check(_lib.lib.ultra_order_trace(trace.data_ptr()))
plan.forward_update(rel, x, w, b, lw, lb, 1e-5, 7, point=point)
torch.cuda.synchronize()
_lib.check(_lib.lib.ultra_order_trace(None))
t = trace.cpu()
main = t[:3 * grid].view(grid, 3).double()
tail = t[3 * grid:7 * grid].view(grid, 4).double()
t0 = main[:, 0].min()
cols = {"start": main[:, 0], "chains done": main[:, 1], "walks done": tail[:, 0], "weights staged": tail[:, 1],
        "wave 0 operands": tail[:, 2], "wave 0 tile done": tail[:, 3], "end": main[:, 2]}
print("ticks since the first workgroup's start (x 10 ns): mean / min / max over %d workgroups" % grid)
prev = None
for name, v in cols.items():
    v = v - t0
    line = "%-18s %8.0f %8.0f %8.0f" % (name, v.mean(), v.min(), v.max())
    if prev is not None:
        line += "   (+%.0f mean since the previous row)" % (v - prev).mean()
    print(line)
    prev = v
