"""Does a full-graph walk of the training step get faster per sample when the batch is walked in slices (smaller working set of the
gathers: the YAGO3-10-shaped x of 8 samples is 252 MB, the last-level cache 256 MB)?  Plan.forward / Plan.backward of the
re-associating plan on bs 8 at once and as 2 x 4, 4 x 2, 8 x 1; HIP events over a hipGraph of the calls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ultra_amd import rspmm, synthetic  # noqa: E402

dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "yago310"
data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False).to(dev)
N, R, bs = data.num_nodes, data.num_relations, 8
plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=False)
g = torch.Generator().manual_seed(0)
rel = torch.randn(bs, R, 64, generator=g).to(dev)
x = torch.randn(bs, N, 64, generator=g).to(dev)
og = torch.randn(bs, N, 64, generator=g).to(dev)
keep = (torch.rand(data.num_edges, generator=g) > 0.001).float().to(dev)


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


for parts in (1, 2, 4, 8):
    step = bs // parts
    out = torch.empty_like(x)

    def fwd():
        for i in range(parts):
            s = slice(i * step, (i + 1) * step)
            plan.forward(rel[s], x[s], edge_weight=keep, keep=True, out=out[s])

    t_f = timed(fwd)
    out_full = plan.forward(rel, x, edge_weight=keep, keep=True)

    def bwd():
        for i in range(parts):
            s = slice(i * step, (i + 1) * step)
            plan.backward(rel[s], x[s], out_full[s], og[s], edge_weight=keep)

    t_b = timed(bwd)
    print("%s  %d x %d samples: forward %.1f us, backward (input + relation gradient) %.1f us" % (shape, parts, step, t_f, t_b), flush=True)
