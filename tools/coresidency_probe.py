"""Do workgroups of two launches share a CU?  Stream A: the entity layer (reference-order kernel, one launch) back to back;
stream B: the relation-graph layer back to back; each alone, then both at once.  With the default build an entity workgroup
(1024 threads, ~152 KB of LDS) owns its CU; the 12-wave build (ULTRA_AMD_LIB=.../libultra_amd_w12.so) leaves four wave slots.
    python tools/coresidency_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import layers, rspmm, synthetic

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234)
N, R = data.num_nodes, int(data.num_relations)
g = torch.Generator().manual_seed(0)
bs = 8
x = torch.randn(bs, N, 64, generator=g).to(dev)
rel = torch.randn(bs, R, 64, generator=g).to(dev)
point = (torch.arange(bs, device=dev) * 7 % N, torch.randn(bs, 64, generator=g).to(dev))
w = (torch.randn(64, 128, generator=g) / 11).to(dev)
b, lw, lb = (torch.randn(64, generator=g).to(dev) for _ in range(3))
plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
layer = layers.GeneralizedRelationalConv(64, 64, 4, 64, "distmult", "sum", True, "relu").to(dev)
rg = data.relation_graph
xr = torch.randn(bs, rg.num_nodes, 64, generator=g).to(dev)
relr = torch.randn(1, 4, 64, generator=g).to(dev).expand(bs, -1, -1)
plan_r = rspmm.Plan(rg.edge_index, rg.edge_type, rg.num_nodes, 4, exact_order=True)
pr = (torch.arange(bs, device=dev), torch.ones(bs, 64, device=dev))


def entity():
    return plan.forward_update(rel, x, w, b, lw, lb, 1e-5, 7, point=point)


def relation():
    return plan_r.fused_layer(relr, xr, layer.linear, layer.layer_norm, residual=True, point=pr)


def graph_of(fn, n, stream):
    fn()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            for _ in range(n):
                fn()
        gr.replay()
    torch.cuda.synchronize()
    return gr


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
NA, NB = 20, 80
ga, gb = graph_of(entity, NA, sa), graph_of(relation, NB, sb)


def timed(run):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def only_a():
    with torch.cuda.stream(sa):
        ga.replay()


def only_b():
    with torch.cuda.stream(sb):
        gb.replay()


def both():
    with torch.cuda.stream(sa):
        ga.replay()
    with torch.cuda.stream(sb):
        gb.replay()


for _ in range(2):
    ta = min(timed(only_a) for _ in range(5))
    tb = min(timed(only_b) for _ in range(5))
    tab = min(timed(both) for _ in range(5))
    print("entity x%d alone %.3f ms (%.1f us each) | relation x%d alone %.3f ms (%.1f us each) | both at once %.3f ms (sum %.3f)"
          % (NA, ta, ta / NA * 1e3, NB, tb, tb / NB * 1e3, tab, ta + tb))
