"""PMC / timing target: the fused relation-graph layer (ultra_nbf_dense_layer) and its unfused pair on the benchmark shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import dense, layers, rspmm, synthetic

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"])
rg = data.relation_graph
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator().manual_seed(0)
N = rg.num_nodes
x = (torch.randn(bs, N, 64, generator=g) / 8).to(dev)
rel = torch.randn(1, 4, 64, generator=g).to(dev).expand(bs, -1, -1)
rows = torch.arange(bs).to(dev)
vals = torch.ones(bs, 64, device=dev)
layer = layers.GeneralizedRelationalConv(64, 64, 4, 64, "distmult", "sum", True, "relu").to(dev)
plan = rspmm.Plan(rg.edge_index, rg.edge_type, N, 4)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    for name, fn in (("fused layer", lambda: plan.fused_layer(rel, x, layer.linear, layer.layer_norm, residual=True, point=(rows, vals))),
                     ("rspmm + update", lambda: dense.conv_update(layer, x, plan.forward(rel, x, point=(rows, vals)), True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("%-16s %.1f us per call (eager launches, bs %d)" % (name, e0.elapsed_time(e1) / iters * 1e3, bs))
