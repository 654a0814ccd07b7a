"""Counter target: the training step's three full-graph walks (forward, input gradient, relation gradient) of the re-associating
plan with a tagged 0/1 keep vector, a few calls each, at a fine-tuning shape (default yago310).  Run under rocprofv3 --pmc."""
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
keep = rspmm.tag_edge_weight((torch.rand(data.num_edges, generator=g) > 0.001).float().to(dev))
for _ in range(4):
    out = plan.forward(rel, x, edge_weight=keep, keep=True)
    plan.backward(rel, x, out, og, edge_weight=keep)
torch.cuda.synchronize()
