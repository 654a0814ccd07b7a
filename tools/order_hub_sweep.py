"""Which hub-row lengths does the reference-order kernel get bit-exact?  python tools/order_hub_sweep.py len..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rspmm_oracle  # noqa: E402
from tests import helpers  # noqa: E402
from ultra_amd.rspmm import Plan  # noqa: E402

dev = torch.device("cuda:0")
for L in [int(a) for a in sys.argv[1:]]:
    case = dict(num_node=300, num_edge=2000, num_relation=9, seed=8, hub=(11, L))
    ei, et = helpers.random_graph(**case)
    N, R, E = 300, 9, ei.shape[1]
    rel, x, w = helpers.features(N, R, 64, E, dtype=torch.float32, seed=8)
    plan = Plan(ei, et, N, R, exact_order=True)
    want = rspmm_oracle.generalized_rspmm(ei, et, torch.ones(E), rel, x, sum="add", mul="mul")
    res = []
    for rep in range(2):
        got = plan.forward(rel.to(dev), x.to(dev), edge_weight=None, sum="add", mul="mul").cpu()
        res.append(int((got != want).any(dim=-1).sum()))
    if res[0] and os.path.isdir("gpurun_out"):
        torch.save(dict(got=got, want=want, ei=ei, et=et), "gpurun_out/hub_%d.pt" % L)
    deg = int(torch.bincount(ei[0], minlength=N)[11])
    print("hub", L, "deg", deg, "chunks", (deg + 59) // 60, "bad rows", res, flush=True)
