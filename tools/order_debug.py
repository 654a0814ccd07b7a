"""Bit-exactness of the reference-order kernel on one tests/test_order_gpu.py case, with the mismatching rows named
(chain row or group item, row length).   python tools/order_debug.py [case index] [sum] [mul]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rspmm_oracle  # noqa: E402
from tests import helpers  # noqa: E402
from tests.test_order_gpu import CASES  # noqa: E402
from ultra_amd.rspmm import Plan  # noqa: E402


def main():
    ci = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    sum_ = sys.argv[2] if len(sys.argv) > 2 else "add"
    mul = sys.argv[3] if len(sys.argv) > 3 else "mul"
    case = CASES[ci]
    dev = torch.device("cuda:0")
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    rel, x, w = helpers.features(N, R, 64, E, dtype=torch.float32, seed=case["seed"])
    plan = Plan(ei, et, N, R, exact_order=True)
    if os.environ.get("UNIT_WALK"):
        from ultra_amd import rspmm
        rspmm.set_tuning(unit_walk=1)
    ones = torch.ones(E)
    want = rspmm_oracle.generalized_rspmm(ei, et, ones, rel, x, sum=sum_, mul=mul)
    deg = torch.bincount(ei[0], minlength=N)
    for rep in range(3):
        got = plan.forward(rel.to(dev), x.to(dev), edge_weight=None, sum=sum_, mul=mul).cpu()
        bad = (got != want).any(dim=-1).nonzero().flatten()
        if len(bad) and os.path.isdir("gpurun_out"):
            torch.save(dict(got=got, want=want, bad=bad), "gpurun_out/order_debug_case%d_rep%d.pt" % (ci, rep))
        print("lib", os.environ.get("ULTRA_AMD_LIB", "default"), "case", ci, "rep", rep, "bad rows", len(bad), "of", N,
              [(int(r), int(deg[r]), int((got[r] != want[r]).sum()), float((got[r] - want[r]).abs().max())) for r in bad[:8]])
        if len(bad):
            r = int(bad[0])
            print("   row", r, "got", got[r][:6].tolist(), "want", want[r][:6].tolist())


if __name__ == "__main__":
    main()
