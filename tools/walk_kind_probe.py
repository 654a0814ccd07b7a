"""The training step's forward walk on the re-associating plan (what it uses) against the reference-order stream walk (aggregate only) with
the same 0/1 keep vector, at the fine-tuning shapes; HIP events over a hipGraph of the calls.  """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ultra_amd import rspmm, synthetic  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, iters=5):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
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
    for _ in range(3):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e3


for shape in sys.argv[1:] or ["fb15k237", "yago310"]:
    data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False).to(dev)
    N, R, bs = data.num_nodes, data.num_relations, 8
    g = torch.Generator().manual_seed(0)
    rel = torch.randn(bs, R, 64, generator=g).to(dev)
    x = torch.randn(bs, N, 64, generator=g).to(dev)
    keep = (torch.rand(data.num_edges, generator=g) > 0.001).float().to(dev)
    keep = rspmm.tag_edge_weight(keep)      # (as the training step does: brought into a plan's edge order once, not once per call)
    out = torch.empty_like(x)
    res = {}
    for kind, exact in (("re-associating", False), ("reference order", True)):
        plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=exact)
        res[kind + ", keep mask"] = timed(lambda: plan.forward(rel, x, edge_weight=keep, keep=True, out=out))
        res[kind + ", no mask"] = timed(lambda: plan.forward(rel, x, out=out))
    print(shape, "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
