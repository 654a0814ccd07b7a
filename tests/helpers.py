"""Shared test utilities: seeded graphs and a host emulation of the kernel's plan walk."""
import numpy as np
import torch

from ultra_amd import _lib

FLT_MAX = {np.float32: np.finfo(np.float32).max, np.float64: np.finfo(np.float64).max}


def random_graph(num_node, num_edge, num_relation, seed=0, hub=None, empty_rows=0, duplicates=0):
    """Unsorted random multigraph.  hub=(node, count) adds a high-degree row; the last `empty_rows`
    node ids never appear as aggregation targets; `duplicates` repeats some (row, col, type) triples."""
    g = torch.Generator().manual_seed(seed)
    hi = max(1, num_node - empty_rows)
    row = torch.randint(0, hi, (num_edge,), generator=g)
    col = torch.randint(0, num_node, (num_edge,), generator=g)
    typ = torch.randint(0, num_relation, (num_edge,), generator=g)
    if hub is not None:
        node, count = hub
        row = torch.cat([row, torch.full((count,), node, dtype=torch.long)])
        col = torch.cat([col, torch.randint(0, num_node, (count,), generator=g)])
        typ = torch.cat([typ, torch.randint(0, num_relation, (count,), generator=g)])
    if duplicates and row.numel():
        idx = torch.randint(0, row.numel(), (duplicates,), generator=g)
        row, col, typ = torch.cat([row, row[idx]]), torch.cat([col, col[idx]]), torch.cat([typ, typ[idx]])
    perm = torch.randperm(row.numel(), generator=g)
    edge_index = torch.stack([row[perm], col[perm]])
    return edge_index, typ[perm]


def features(num_node, num_relation, dim, num_edge, dtype=torch.float32, seed=1, unit_weight=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(num_node, dim, generator=g, dtype=torch.float64).to(dtype)
    rel = torch.randn(num_relation, dim, generator=g, dtype=torch.float64).to(dtype)
    w = torch.ones(num_edge, dtype=dtype) if unit_weight else (torch.rand(num_edge, generator=g, dtype=torch.float64) + 0.5).to(dtype)
    return rel, x, w


def _nary(sum, a, b):
    if sum == "add":
        return a + b
    if sum == "min":
        return np.where(a < b, a, b)
    return np.where(a > b, a, b)


def _zero(sum, dtype):
    if sum == "add":
        return dtype(0)
    return FLT_MAX[dtype] if sum == "min" else -FLT_MAX[dtype]


def emulate_plan_forward(plan, relation, input, edge_weight=None, boundary=None, sum="add", mul="mul"):
    """Walk the exported plan exactly like rspmm_fwd_kernel does (same grouping, same operation
    order, separately rounded products and sums) with numpy.  Slow: small graphs only."""
    typed = getattr(plan, "typed", None) is not None and sum == "add" and mul == "mul"
    if typed:
        plan = plan.typed       # the type-run twin serves add_mul: items hold one relation each
    info = plan.info()
    col = plan.export(_lib.ARR_COL).numpy()
    typ = plan.export(_lib.ARR_TYPE).numpy()
    perm = plan.export(_lib.ARR_PERM).numpy()
    items = plan.export(_lib.ARR_ITEM).numpy().reshape(-1, 4)
    split_row = plan.export(_lib.ARR_SPLIT_ROW).numpy()
    split_ptr = plan.export(_lib.ARR_SPLIT_PTR).numpy()
    rel = relation.numpy()
    x = input.numpy()
    dtype = x.dtype.type
    D = x.shape[1]
    w_sorted = None if edge_weight is None else edge_weight.numpy()[perm]
    out = np.empty((info["num_node"], D), dtype=dtype)
    partial = np.empty((max(info["n_partial_slot"], 1), D), dtype=dtype)
    n_w = info["n_wave_item"]

    def msg(k):
        r, xi = rel[typ[k]], x[col[k]]
        y = xi if typed else (r * xi if mul == "mul" else r + xi)
        if w_sorted is not None:
            y = w_sorted[k] * y
        return y.astype(dtype)

    def walk(begin, cnt, stride):
        acc = np.full(D, _zero(sum, dtype), dtype=dtype)
        for s in range(cnt):
            acc = _nary(sum, acc, msg(begin + s * stride)).astype(dtype)
        return acc

    def emit(row, slot, acc):
        if slot >= 0:
            partial[slot] = acc
        else:
            if boundary is not None:
                acc = _nary(sum, acc, boundary.numpy()[row]).astype(dtype)
            out[row] = acc

    seen_rows = set()
    for idx, (row, begin, length, slot) in enumerate(items):
        if idx < n_w:   # wave item: four strided groups, then (g0+g1)+(g2+g3)
            g = [walk(begin + q, max(0, (length - q + 3) >> 2), 4) for q in range(4)]
            acc = _nary(sum, _nary(sum, g[0], g[1]).astype(dtype), _nary(sum, g[2], g[3]).astype(dtype)).astype(dtype)
        else:
            acc = walk(begin, length, 1)
        if typed and length > 0:        # rel[type] (x) sum of the run, applied once
            acc = (rel[typ[begin]] * acc).astype(dtype)
        elif typed:
            acc = (rel[0] * acc).astype(dtype)
        emit(row, slot, acc)
        seen_rows.add(int(row))
    for k, row in enumerate(split_row):
        # rspmm_fixup_kernel: 16 slot lanes sum the partials s, s + 16, ... in order; the lane sums are folded 0 .. 15
        lanes = []
        for s in range(16):
            acc = np.full(D, _zero(sum, dtype), dtype=dtype)
            for sl in range(split_ptr[k] + s, split_ptr[k + 1], 16):
                acc = _nary(sum, acc, partial[sl]).astype(dtype)
            lanes.append(acc)
        acc = lanes[0]
        for s in range(1, 16):
            acc = _nary(sum, acc, lanes[s]).astype(dtype)
        if boundary is not None:
            acc = _nary(sum, acc, boundary.numpy()[row]).astype(dtype)
        out[row] = acc
    assert len(seen_rows) == info["num_node"], "every output row must be owned by at least one item"
    return torch.from_numpy(out)


def assert_sum_close(got, want, edge_index, edge_type, edge_weight, relation, input, mul="mul", boundary=None, k=None):
    """Two fp sums of the same n terms in different association orders differ by a random walk of
    roundings, each <= eps * |partial sum| <= eps * sum|terms|: bound = (2 + sqrt(n)) * eps * sum|terms|
    per element, a data-dependent tolerance instead of a blanket one."""
    if k is None:
        deg = torch.bincount(edge_index[0].cpu(), minlength=1).max().item() if edge_index.shape[1] else 0
        k = 2.0 + float(deg) ** 0.5
    from oracle import rspmm_oracle
    mass = rspmm_oracle.generalized_rspmm(edge_index, edge_type, edge_weight.abs(), relation.abs(), input.abs(),
                                          sum="add", mul=mul)
    if boundary is not None:
        mass = mass + boundary.abs()
    eps = torch.finfo(want.dtype).eps
    bound = k * eps * mass + 10 * torch.finfo(want.dtype).tiny
    diff = (got - want).abs()
    bad = diff > bound
    assert not bad.any(), "max excess %g at %s" % ((diff - bound).max().item(), bad.nonzero()[0].tolist())
