"""Chain phase of the reference-order kernel: cycles until a workgroup's chains are done against the number of 60-edge chunks
it was dealt (sorted against sorted: the workgroup -> partition mapping is left out), least-squares slope = cycles per chunk,
intercept = what precedes the chains (relation slice staging).

    python tools/chain_probe.py [shape] [batch]
"""
import ctypes
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
nparts = grid // bs
plan.forward(rel, x, point=point)
torch.cuda.synchronize()
trace = torch.zeros(grid * 32, dtype=torch.int64, device=dev)
_lib.check(_lib.lib.ultra_order_trace(trace.data_ptr()))
plan.forward(rel, x, point=point)
torch.cuda.synchronize()
_lib.check(_lib.lib.ultra_order_trace(None))
t = trace.cpu()[:3 * grid].view(grid, 3).double()
staged = trace.cpu()[7 * grid:8 * grid].double() - t[:, 0]
print("relation slice staged (cycles since the workgroup's start): min %.0f mean %.0f max %.0f" % (staged.min(), staged.mean(), staged.max()))
t0 = t[:, 0]
print("workgroup starts, relative to the first: mean %.0f max %.0f" % ((t0 - t0.min()).mean(), (t0 - t0.min()).max()))
chain = (t[:, 1] - t[:, 0]).sort()[0]
end = (t[:, 2] - t[:, 0])
n = ctypes.c_int64()
_lib.check(_lib.lib.ultra_plan_schedule_export(plan._h, nparts, 0, None, 0, ctypes.byref(n)))
cp = torch.empty(n.value, dtype=torch.int32)
_lib.check(_lib.lib.ultra_plan_schedule_export(plan._h, nparts, 0, cp.data_ptr(), n.value, ctypes.byref(n)))
chunks = (cp[1:] - cp[:-1]).double()
per_wg = chunks.repeat(bs).sort()[0]
A = torch.stack([per_wg, torch.ones_like(per_wg)], dim=1)
sol = torch.linalg.lstsq(A, chain.unsqueeze(1)).solution.flatten()
print("%s bs %d: %d partitions, chunks per partition min %d mean %.1f max %d" % (shape, bs, nparts, chunks.min(), chunks.mean(), chunks.max()))
print("chains done (cycles since the workgroup's start): min %.0f mean %.0f max %.0f; kernel end mean %.0f max %.0f" %
      (chain.min(), chain.mean(), chain.max(), end.mean(), end.max()))
print("least squares: %.0f cycles per chunk + %.0f cycles before the chains" % (sol[0], sol[1]))
for q in (0, 64, 128, 192, 255):
    print("  workgroup rank %3d: %5.0f chunks, chains done at %7.0f" % (q, per_wg[q], chain[q]))
