"""Minimal PMC target: the two rspmm kernels of the benchmark forward (FB15k237-shaped, batch 8,
add_mul + fused boundary), 1 warm-up + 3 launches each.  Run under rocprofv3 --pmc <counters>."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import rspmm, synthetic  # noqa: E402

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234)
bs = 8
g = torch.Generator().manual_seed(0)
for graph, R in ((data, data.num_relations), (data.relation_graph, 4)):
    N = graph.num_nodes
    x = torch.randn(bs, N, 64, generator=g).to(dev)
    rel = torch.randn(bs, R, 64, generator=g).to(dev)
    bnd = torch.randn(bs, N, 64, generator=g).to(dev)
    plan = rspmm.Plan(graph.edge_index, graph.edge_type, N, R)
    point = (torch.arange(bs, device=dev), bnd[:, 0].contiguous())
    ms, _ = plan.forward_timed(rel, x, point=point, warmup=1, iters=3)
    print("N=%d ms=%.4f" % (N, ms))
    del plan
