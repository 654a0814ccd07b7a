"""Debugging aid for the form-3 update waves (build with -DULTRA_SPIN_GUARD=1, tools/build_variant.py): run one launch on a small
graph with a chain row and print where a spin gave up.

    python tools/build_variant.py guard -DULTRA_SPIN_GUARD=1
    ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_guard.so python tools/spin_guard_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers  # noqa: E402
from ultra_amd import _lib, dense, rspmm  # noqa: E402

dev = torch.device("cuda:0")
case = dict(num_node=64, num_edge=300, num_relation=3, seed=1, hub=(7, 700))
ei, et = helpers.random_graph(**case)
N, R, bs = case["num_node"], case["num_relation"], 8
plan = rspmm.Plan(ei, et, N, R, exact_order=True)
g = torch.Generator().manual_seed(0)
x = torch.randn(bs, N, 64, generator=g).to(dev)
rel = torch.randn(bs, R, 64, generator=g).to(dev)
w = (torch.randn(64, 128, generator=g) / 11).to(dev)
b, lw, lb = (torch.randn(64, generator=g).to(dev) for _ in range(3))
grid = 256
trace = torch.zeros(grid * 64, dtype=torch.int64, device=dev)
_lib.check(_lib.lib.ultra_order_trace(trace.data_ptr()))
rspmm.set_tuning(update_form=3)
got = plan.forward_update(rel, x, w, b, lw, lb, 1e-5, 7, point=None, sum="add")
torch.cuda.synchronize()
_lib.check(_lib.lib.ultra_order_trace(None))
rspmm.set_tuning()
want = dense._conv_update_forward(x, plan.forward(rel, x, sum="add", mul="mul"), w, b, lw, lb, 1e-5, 7)
print("served", got is not None, "equal", got is not None and torch.equal(got, want))
t = trace.cpu()[24 * grid:40 * grid].view(grid, 4, 4)
n = 0
for wg in range(grid):
    for u in range(4):
        a, b_, c, d = (int(v) for v in t[wg, u])
        if a:
            n += 1
            if n <= 24:
                print("workgroup %d update wave %d: spin %d gave up at tile %d, epoch %d | tail %d walked %d | barrier word %d consumed %d | "
                      "posted %d %d" % (wg, u, a >> 48, (a >> 32) & 0xffff, a & 0xffffffff, b_ >> 32, b_ & 0xffffffff, c >> 32,
                                        c & 0xffffffff, d >> 32, d & 0xffffffff))
print("%d reports" % n)
