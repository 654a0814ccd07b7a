"""Relation-graph layer in reference order (ultra_nbf_dense_layer) on the FB15k237-shaped relation graph: time per launch
(back-to-back launches from Python: a floor of ~14 us per call is launch overhead, not kernel time)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import layers, rspmm, synthetic
dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234)
rg = data.relation_graph
plan = rspmm.Plan(rg.edge_index, rg.edge_type, rg.num_nodes, 4, exact_order=True)
layer = layers.GeneralizedRelationalConv(64, 64, 4, 64, "distmult", "sum", True, "relu").to(dev)
x = torch.randn(8, rg.num_nodes, 64, device=dev)
rel = torch.randn(8, 4, 64, device=dev)
run = lambda: plan.fused_layer(rel, x, layer.linear, layer.layer_norm, relu=True, residual=True)
if os.environ.get("PROBE_GRID"):       # a launch-grid tuning below the CU count: the two-tile form of the layer
    rspmm.set_tuning(grid=int(os.environ["PROBE_GRID"]))
for _ in range(5):
    run()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay()
e0.record()
for _ in range(5):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("dense_order_layer, graph of 20 launches: %.1f us per launch" % (e0.elapsed_time(e1) / 100 * 1e3))
