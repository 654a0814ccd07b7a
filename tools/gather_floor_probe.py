"""What bounds the entity-graph rspmm kernel?  Same plan structure (rows, degrees, types), but every gathered source
row folded onto a few rows (col % K): the gathers then hit L1 / L2 trivially and what remains is the kernel's
non-gather floor (record stream, LDS relation reads, VALU, stores)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import rspmm, synthetic

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], relation_graph=False)
N, R, bs = data.num_nodes, data.num_relations, 8
g = torch.Generator().manual_seed(0)
x = torch.randn(bs, N, 64, generator=g).to(dev)
rel = torch.randn(bs, R, 64, generator=g).to(dev)
point = (torch.arange(bs, device=dev), torch.randn(bs, 64, generator=g).to(dev))
for name, K in (("original columns", None), ("col % 4096", 4096), ("col % 128", 128), ("col % 8", 8)):
    ei = data.edge_index.clone()
    if K:
        ei[1] = ei[1] % K
    plan = rspmm.Plan(ei, data.edge_type, N, R)
    ms, _ = plan.forward_timed(rel, x, point=point, warmup=5, iters=50)
    print("%-18s main kernel %.1f us   (call incl. fix-up %.1f us)" % (name, plan.last_main_kernel_ms * 1e3, ms * 1e3))
