"""Time the relation-graph rspmm of the benchmark shape: dense-format plan vs type-run plan (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import rspmm, synthetic

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"])
rg = data.relation_graph
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator().manual_seed(0)
x = torch.randn(bs, rg.num_nodes, 64, generator=g).to(dev)
rel = torch.randn(1, 4, 64, generator=g).to(dev).expand(bs, -1, -1)
print("relation graph: N=%d E=%d fill=%.3f" % (rg.num_nodes, rg.num_edges, rg.num_edges / (rg.num_nodes ** 2 * 4.0)))
for name, kw in (("dense", dict(dense=True, type_runs=False)), ("typed", dict(dense=False, type_runs=True)),
                 ("rowcol", dict(dense=False, type_runs=False))):
    plan = rspmm.Plan(rg.edge_index, rg.edge_type, rg.num_nodes, 4, **kw)
    ms, out = plan.forward_timed(rel, x, boundary=x, warmup=5, iters=50)
    print("%-7s %.4f ms/call  main kernel %.4f ms" % (name, ms, plan.last_main_kernel_ms))
    if name == "dense":
        ref = out.clone()
    else:
        print("   max |d| vs dense = %.3e (max |out| %.3e)" % ((out - ref).abs().max().item(), ref.abs().max().item()))
