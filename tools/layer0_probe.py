"""Time ultra_nbf_layer0 (fill + rows kernels) on the benchmark graphs for ordinary and hub source rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import rspmm, synthetic

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"]).to(dev)
bs = 8
lin = torch.nn.Linear(128, 64).to(dev)
ln = torch.nn.LayerNorm(64).to(dev)
g = torch.Generator().manual_seed(0)


def timed(fn, iters=20):
    """GPU time per call: the calls are captured into one hipGraph, so host launch overhead does not count."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


for name, graph, R in (("entity", data, data.num_relations), ("relation", data.relation_graph, 4)):
    N = graph.num_nodes
    plan = rspmm.get_plan(graph.edge_index, graph.edge_type, N, R)
    rel = torch.randn(bs, R, 64, generator=g).to(dev)
    vals = torch.randn(bs, 64, generator=g).to(dev)
    deg = torch.bincount(graph.edge_index[1], minlength=N)
    order = deg.argsort(descending=True)
    picks = {"hub rows": order[:bs], "median rows": order[N // 2:N // 2 + bs], "test heads": data.target_triples[:bs, 0] if name == "entity" else data.target_triples[:bs, 2]}
    for what, rows in picks.items():
        rows = rows.contiguous()
        with torch.no_grad():
            us = timed(lambda: plan.layer0(rel, rows, vals, lin, ln, relu=True, residual=True))
        print("%-8s %-12s out-degrees %s -> %.1f us (fill + rows)" % (name, what, deg[rows].tolist(), us))
