"""Entity-graph rspmm (add_mul, point boundary): the default plan against the reference-order plan
(ULTRA_PLAN_EXACT_ORDER), HIP-event time per call.  python tools/exact_order_probe.py [shape bs]..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import rspmm, synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    points = [("fb15k237", 8), ("codex_l", 8)]
    if len(sys.argv) > 2:
        points = [(sys.argv[i], int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    for shape, bs in points:
        data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
        N, R = data.num_nodes, data.num_relations
        g = torch.Generator().manual_seed(0)
        x = torch.randn(bs, N, 64, generator=g).to(dev)
        rel = torch.randn(bs, R, 64, generator=g).to(dev)
        point = (torch.arange(bs, device=dev) * 7 % N, torch.randn(bs, 64, generator=g).to(dev))
        deg = torch.bincount(data.edge_index[0], minlength=N)
        print("%s bs=%d: max degree %d, rows > 256: %d holding %d of %d edges" %
              (shape, bs, deg.max(), (deg > 256).sum(), deg[deg > 256].sum(), data.num_edges), flush=True)
        for exact in (False, True):
            plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=exact)
            ms, out = plan.forward_timed(rel, x, point=point, warmup=3, iters=20)
            print("  exact_order=%s: %.3f ms per call (main kernel %.3f ms)" % (exact, ms, plan.last_main_kernel_ms), flush=True)
            del plan


if __name__ == "__main__":
    main()
