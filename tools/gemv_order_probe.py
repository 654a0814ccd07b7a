"""Prints the summation tree the host BLAS uses for y = x @ w.T with ONE output column (nn.Linear(K, 1): the readout's last
product, models.py:208), the stage program the readout kernel would run for it, and how much of torch it reproduces.
CPU only (ultra_amd/host_order.py does the probing).

    python tools/gemv_order_probe.py [K=128] [out.json]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import host_order as ho  # noqa: E402


def show(t):
    return str(t) if isinstance(t, int) else "(" + show(t[0]) + " " + show(t[1]) + ")"


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    print(torch.__config__.parallel_info().splitlines()[1].strip(), file=sys.stderr)
    tree = ho.probe_tree(K)
    print(show(tree))
    if len(sys.argv) > 2:
        json.dump(tree, open(sys.argv[2], "w"))
    try:
        stages = ho.tree_to_stages(ho.annotate(tree, K))
    except ValueError as exc:
        sys.exit("outside the lanes-and-fold family: %s" % exc)
    for L, carry, lists in stages:
        print("stage: %d lanes%s, elements per lane %s, unfused steps %d"
              % (L, " (lane 0 continues from the previous stage)" if carry else "", [len(l) for l in lists],
                 sum(1 for l in lists for k in l if k & ho.UNFUSED)))
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(8192, K, generator=g), torch.randn(1, K, generator=g)
    same = (ho.emulate(stages, x.numpy(), w[0].numpy()) == torch.nn.functional.linear(x, w)[:, 0].numpy()).mean()
    print("program vs torch on 8192 random rows: %.3f %% bit-equal" % (100 * same))
