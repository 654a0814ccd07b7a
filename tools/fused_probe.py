"""Layer time at the benchmark point (FB15k237 shape, bs 8): rspmm + conv_update as two launches vs the one launch whose tail
applies the update (ultra_rspmm_forward_update).  argv: [shape] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import dense, rspmm, synthetic

dev = torch.device("cuda:0")
shape = sys.argv[1] if len(sys.argv) > 1 else "fb15k237"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
ei, et = data.edge_index.to(dev), data.edge_type.to(dev)
N, R = data.num_nodes, int(data.num_relations)
plan = rspmm.get_plan(ei, et, N, R)
g = torch.Generator().manual_seed(0)
rel = torch.randn(bs, R, 64, generator=g).to(dev)
x = torch.randn(bs, N, 64, generator=g).to(dev)
rows = torch.randint(0, N, (bs,), generator=g).to(dev)
vals = torch.randn(bs, 64, generator=g).to(dev)
w = (torch.randn(64, 128, generator=g) / 11).to(dev)
b, lw, lb = (torch.randn(64, generator=g).to(dev) for _ in range(3))


def two():
    agg = plan.forward(rel, x, sum="add", mul="mul", point=(rows, vals))
    return dense._conv_update_forward(x, agg, w, b, lw, lb, 1e-5, 7)


def one():
    return plan.forward_update(rel, x, w, b, lw, lb, 1e-5, 7, point=(rows, vals))


def one_no_matrix():   # measurement flag 256: the tail without its matrix chain (wrong results)
    return plan.forward_update(rel, x, w, b, lw, lb, 1e-5, 7 | 256, point=(rows, vals))


def agg_only():
    return plan.forward(rel, x, sum="add", mul="mul", point=(rows, vals))


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("%s bs %d: bit-equal %s" % (shape, bs, torch.equal(one(), two())))
for k in range(2):
    print("aggregate only %.1f us | two launches %.1f us | one launch %.1f us | one launch without the matrix chain %.1f us"
          % (timed(agg_only), timed(two), timed(one), timed(one_no_matrix)))
