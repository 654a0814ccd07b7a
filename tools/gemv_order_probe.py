"""Recovers the summation tree the host BLAS uses for y = x @ w.T with ONE output column (nn.Linear(K, 1): the readout's
last product, models.py:208), by probing: a row with x_i = 1, x_j = -1, x_k = 2^-30 (everything else 0, w = 1) gives
2^-30 iff i and j are added together before k joins them -- (1 + 2^-30) rounds to 1 in fp32.  O(K^2) probes per tree level,
one F.linear call per 8192 probes.  Prints the tree as nested pairs and writes it as JSON.

    python tools/gemv_order_probe.py [K=128] [rows per call=8192] [out.json]
"""
import json
import sys

import torch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
MROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
OUT = sys.argv[3] if len(sys.argv) > 3 else "gemv_tree.json"
w = torch.ones(1, K)
TINY = 2.0 ** -30


def run_queries(qs):
    out = []
    for s in range(0, len(qs), MROWS):
        chunk = qs[s:s + MROWS]
        x = torch.zeros(MROWS, K)
        for r, (i, j, k) in enumerate(chunk):
            x[r, i], x[r, j], x[r, k] = 1.0, -1.0, TINY
        y = torch.nn.functional.linear(x, w)[:, 0]
        out += [bool(v != 0) for v in y[:len(chunk)].tolist()]
    return out


def solve(leaves):
    if len(leaves) == 1:
        return leaves[0]
    if len(leaves) == 2:
        return (leaves[0], leaves[1])
    a, rest = leaves[0], leaves[1:]
    qs = [(a, j, k) for j in rest for k in rest if j != k]
    res = iter(run_queries(qs))
    below = {(j, k): next(res) for j in rest for k in rest if j != k}          # LCA(a, j) strictly below LCA(a, k)
    level = {j: sum(1 for k in rest if k != j and below[(k, j)]) for j in rest}
    groups = {}
    for j in rest:
        groups.setdefault(level[j], []).append(j)
    node = a
    for lv in sorted(groups):
        node = (node, solve(groups[lv]))
    return node


def show(t):
    return str(t) if isinstance(t, int) else "(" + show(t[0]) + " " + show(t[1]) + ")"


if __name__ == "__main__":
    print(torch.__config__.parallel_info().splitlines()[0:3], file=sys.stderr)
    tree = solve(list(range(K)))
    print(show(tree))
    json.dump(tree, open(OUT, "w"))
