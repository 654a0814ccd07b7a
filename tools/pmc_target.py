"""Minimal PMC target: the kernels that make up 90 % of the benchmark forward (FB15k237 shape, batch 8), in the reference's
operation order -- entity-graph rspmm (add_mul, point boundary), the entity-graph layer update, the two in one launch (what the step
runs), the relation-graph layer, the readout.
1 warm-up + a few launches each.  Run under rocprofv3 --pmc <counters> (tools/collect_profiles.sh C)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import dense, layers, models, rspmm, synthetic  # noqa: E402

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234)
bs = 8
g = torch.Generator().manual_seed(0)
layer = layers.GeneralizedRelationalConv(64, 64, 4, 64, "distmult", "sum", True, "relu").to(dev)
with torch.no_grad():
    # entity graph: rspmm main kernel (+ fix-up), then the update kernel
    N, R = data.num_nodes, data.num_relations
    x = torch.randn(bs, N, 64, generator=g).to(dev)
    rel = torch.randn(bs, R, 64, generator=g).to(dev)
    point = (torch.arange(bs, device=dev) * 7 % N, torch.randn(bs, 64, generator=g).to(dev))
    plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
    for _ in range(4):
        agg = plan.forward(rel, x, point=point)
    for _ in range(4):
        dense.conv_update(layer, x, agg, True)
    # ... and the two as the step runs them: one launch (ultra_rspmm_forward_update)
    ln = layer.layer_norm
    for _ in range(4):
        one = plan.forward_update(rel, x, layer.linear.weight, layer.linear.bias, ln.weight, ln.bias, ln.eps, 7, point=point)
    assert one is not None
    # relation graph: the whole layer in one launch
    rg = data.relation_graph
    xr = torch.randn(bs, rg.num_nodes, 64, generator=g).to(dev)
    relr = torch.randn(1, 4, 64, generator=g).to(dev).expand(bs, -1, -1)
    plan_r = rspmm.Plan(rg.edge_index, rg.edge_type, rg.num_nodes, 4, exact_order=True)
    pr = (torch.arange(bs, device=dev), torch.ones(bs, 64, device=dev))
    for _ in range(4):
        out = plan_r.fused_layer(relr, xr, layer.linear, layer.layer_norm, residual=True, point=pr)
    # readout over all candidates
    net = models.EntityNBFNet(**{k: v for k, v in synthetic.default_model_cfg()["entity_model_cfg"].items() if k != "class"}).to(dev)
    query = torch.randn(bs, 64, generator=g).to(dev)
    every = torch.arange(N, device=dev).unsqueeze(0).expand(bs, -1).contiguous()
    for _ in range(4):
        score = dense.readout(net, x, query, every)
    torch.cuda.synchronize()
    print("ok", float(agg.abs().mean()), float(out.abs().mean()), float(score.abs().mean()))
