"""Round 6's new kernels and routes, each against the ORACLE directly (oracle/rspmm_oracle.py: the C restatement of the
reference's rspmm.cpp, pinned to the reference translation unit by tests/test_oracle.py) -- not against a sibling route."""
import pytest
import torch

from oracle import rspmm_oracle
from ultra_amd import rspmm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle_backward(ei, et, w, rel, x, og, mul="mul"):
    """rspmm_backward_cpu on sorted edges, sample by sample: (relation_grad, input_grad) in the batch-major layout."""
    sei, set_, sw, order = rspmm_oracle.sort_edges(ei, et, w)
    rgs, xgs = [], []
    for b in range(x.shape[0]):
        out = rspmm_oracle.rspmm_forward(sei, set_, sw, rel[b], x[b], sum="add", mul=mul)
        _, rg, xg = rspmm_oracle.rspmm_backward(sei, set_, sw, rel[b], x[b], out, og[b], sum="add", mul=mul)
        rgs.append(rg)
        xgs.append(xg)
    return torch.stack(rgs), torch.stack(xgs)


def _dense_graph(n, num_type, fill, seed):
    """A relation-graph-like edge list: every (row, type, col) cell holds an edge with probability `fill`."""
    gen = torch.Generator().manual_seed(seed)
    cell = torch.rand(num_type, n, n, generator=gen) < fill
    t, r, c = cell.nonzero().t()
    return torch.stack([r, c]), t


@pytest.mark.parametrize("n,num_type,bs", [(474, 4, 8), (100, 4, 3), (70, 3, 2), (33, 1, 1)])
def test_dense_relation_gradient_against_the_oracle(dev, n, num_type, bs):
    """ultra_rspmm_dense_relation_grad (the relation graph's backward on the matrix cores) vs rspmm_backward_cpu's relation_grad;
    the input gradient of the same call (dense transposed twin) rides along."""
    ei, et = _dense_graph(n, num_type, 0.6, seed=n)
    gen = torch.Generator().manual_seed(1)
    rel = torch.randn(bs, num_type, 64, generator=gen)
    x = torch.randn(bs, n, 64, generator=gen)
    og = torch.randn(bs, n, 64, generator=gen)
    w = torch.ones(ei.shape[1])
    want_rg, want_xg = _oracle_backward(ei, et, w, rel, x, og)
    plan = rspmm.Plan(ei, et, n, num_type, exact_order=False)
    assert plan.dense is not None
    out = plan.forward(rel.to(dev), x.to(dev))
    _, rg, xg = plan.backward(rel.to(dev), x.to(dev), out, og.to(dev))
    # every sum here has hundreds to 10^5 terms of magnitude ~1: compare at fp32 accumulation accuracy of that length
    scale_r = want_rg.abs().max().item()
    assert (rg.cpu() - want_rg).abs().max().item() <= 3e-5 * scale_r
    assert (xg.cpu() - want_xg).abs().max().item() <= 3e-5 * want_xg.abs().max().item()
    # the switch selects the edge walk (what round 5 ran): same gradient to rounding
    rspmm.DENSE_RELATION_GRAD = False
    try:
        _, rg_walk, _ = plan.backward(rel.to(dev), x.to(dev), out, og.to(dev))
    finally:
        rspmm.DENSE_RELATION_GRAD = True
    assert (rg_walk.cpu() - want_rg).abs().max().item() <= 3e-5 * scale_r
    # deterministic: no atomics on this route
    _, rg2, _ = plan.backward(rel.to(dev), x.to(dev), out, og.to(dev))
    assert torch.equal(rg, rg2)


# ---- rspmm on a list of output rows: forward, scatter backward (round 5) and gather backward (round 6) vs the oracle ----
def _rows_case(seed, n=500, e=7000, bs=3, num_rel=7, n_list=40, masked=True):
    gen = torch.Generator().manual_seed(seed)
    ei = torch.randint(1, n, (2, e), generator=gen)
    ei[0, :1800] = 4                                      # a hub row on the aggregation side
    ei[1, 1800:2300] = 9                                  # a hub source (several segments of its out-edge list)
    et = torch.randint(0, num_rel, (e,), generator=gen)
    et[:3000] = 1                                         # a common relation type
    rows = torch.randint(1, n, (bs, n_list), generator=gen)
    rows[:, 0] = 4                                        # the hub
    rows[:, 1] = rows[:, 2]                               # a repeated entry
    rows[:, 3] = 0                                        # a row without in-edges
    rows[:, 4] = 9
    point_rows = torch.tensor([4, 17, 0][:bs])
    if bs > 1:
        rows[1, 5] = 17                                   # a listed row that carries the boundary value
    keep = (torch.rand(e, generator=gen) > 0.25).float() if masked else torch.ones(e)
    rel = torch.randn(bs, num_rel, 64, generator=gen)
    x = torch.randn(bs, n, 64, generator=gen)
    values = torch.randn(bs, 64, generator=gen)
    gagg = torch.randn(bs, n_list, 64, generator=gen)
    gupd = torch.randn(bs, n_list, 64, generator=gen)
    return ei, et, rows, point_rows, keep, rel, x, values, gagg, gupd


def _call_rows_forward(plan, mul, keep, rel, x, rows, point_rows, values):
    import ctypes
    from ultra_amd._lib import MUL_CODES, check, lib
    _, mrel = rspmm.as_mat(rel)
    _, mx = rspmm.as_mat(x)
    agg = torch.empty(rows.shape[0], rows.shape[1], 64, device=x.device)
    check(lib.ultra_rspmm_rows_forward(plan._h, MUL_CODES[mul], keep.data_ptr() if keep is not None else None, ctypes.byref(mrel),
                                       ctypes.byref(mx), rows.data_ptr(), rows.shape[1], None,
                                       point_rows.data_ptr() if point_rows is not None else None,
                                       values.data_ptr() if values is not None else None, agg.data_ptr(),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return agg


@pytest.mark.parametrize("mul", ["mul", "add"])
@pytest.mark.parametrize("masked", [True, False])
def test_rows_forward_against_the_oracle(dev, mul, masked):
    """ultra_rspmm_rows_forward == rspmm_forward_cpu(...)[rows] (+ the point boundary's value on the query's row)."""
    ei, et, rows, point_rows, keep, rel, x, values, _, _ = _rows_case(51, masked=masked)
    sei, set_, sw, _ = rspmm_oracle.sort_edges(ei, et, keep)
    plan = rspmm.Plan(ei, et, x.shape[1], rel.shape[1], exact_order=False)
    got = _call_rows_forward(plan, mul, keep.to(dev), rel.to(dev), x.to(dev), rows.to(dev), point_rows.to(dev), values.to(dev)).cpu()
    for b in range(x.shape[0]):
        full = rspmm_oracle.rspmm_forward(sei, set_, sw, rel[b], x[b], sum="add", mul=mul)
        full[point_rows[b]] += values[b]
        want = full[rows[b]]
        assert (got[b] - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("mul", ["mul", "add"])
@pytest.mark.parametrize("route", ["gather", "scatter"])
def test_rows_backward_against_the_oracle(dev, mul, route):
    """Both backward entries of the listed-rows rspmm vs rspmm_backward_cpu with an output_grad that is zero off the listed rows
    (repeated entries add up): relation_grad and input_grad; the gather entry also adds the update's share on the listed rows,
    returns the point boundary's gradient, and is reproducible bit for bit."""
    import ctypes
    from ultra_amd._lib import MUL_CODES, check, lib
    ei, et, rows, point_rows, keep, rel, x, values, gagg, gupd = _rows_case(52)
    bs, n = x.shape[:2]
    sei, set_, sw, _ = rspmm_oracle.sort_edges(ei, et, keep)
    want_rg, want_xg, want_vg = [], [], []
    for b in range(bs):
        og = torch.zeros(n, 64).index_add_(0, rows[b], gagg[b])
        out = rspmm_oracle.rspmm_forward(sei, set_, sw, rel[b], x[b], sum="add", mul=mul)
        _, rg, xg = rspmm_oracle.rspmm_backward(sei, set_, sw, rel[b], x[b], out, og, sum="add", mul=mul)
        if route == "gather":
            xg = xg + torch.zeros(n, 64).index_add_(0, rows[b], gupd[b])
        want_rg.append(rg), want_xg.append(xg), want_vg.append(og[point_rows[b]])
    want_rg, want_xg, want_vg = torch.stack(want_rg), torch.stack(want_xg), torch.stack(want_vg)

    plan = rspmm.Plan(ei, et, n, rel.shape[1], exact_order=False)
    d = lambda t: t.to(dev).contiguous()
    keep_d, rel_d, x_d, rows_d, pr_d, gagg_d, gupd_d = d(keep), d(rel), d(x), d(rows), d(point_rows), d(gagg), d(gupd)
    _, mrel = rspmm.as_mat(rel_d)
    _, mx = rspmm.as_mat(x_d)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        if route == "gather":
            rg, xg = torch.full_like(rel_d, float("nan")), torch.full_like(x_d, float("nan"))      # written in full
            vg = torch.full((bs, 64), float("nan"), device=dev)
        else:
            rg, xg, vg = torch.zeros_like(rel_d), torch.zeros_like(x_d), None
        _, mrg = rspmm.as_mat(rg)
        _, mxg = rspmm.as_mat(xg)
        if route == "gather":
            check(lib.ultra_rspmm_rows_backward_gather(plan._h, MUL_CODES[mul], keep_d.data_ptr(), ctypes.byref(mrel), ctypes.byref(mx),
                                                       rows_d.data_ptr(), rows.shape[1], gagg_d.data_ptr(), gupd_d.data_ptr(),
                                                       pr_d.data_ptr(), vg.data_ptr(), ctypes.byref(mrg), ctypes.byref(mxg), stream))
        else:
            check(lib.ultra_rspmm_rows_backward(plan._h, MUL_CODES[mul], keep_d.data_ptr(), ctypes.byref(mrel), ctypes.byref(mx),
                                                rows_d.data_ptr(), rows.shape[1], gagg_d.data_ptr(), ctypes.byref(mrg),
                                                ctypes.byref(mxg), stream))
        return rg, xg, vg

    rg, xg, vg = run()
    assert (rg.cpu() - want_rg).abs().max().item() <= 3e-5 * max(1.0, want_rg.abs().max().item())
    assert (xg.cpu() - want_xg).abs().max().item() <= 3e-5 * max(1.0, want_xg.abs().max().item())
    if route == "gather":
        assert (vg.cpu() - want_vg).abs().max().item() <= 1e-5 * max(1.0, want_vg.abs().max().item())
        rg2, xg2, vg2 = run()
        assert torch.equal(rg, rg2) and torch.equal(xg, xg2) and torch.equal(vg, vg2)


def test_rows_backward_gather_serves_more_than_eight_samples_and_declines_long_lists(dev):
    import ctypes
    from ultra_amd._lib import ULTRA_ERR_UNSUPPORTED, lib
    ei, et, rows, point_rows, keep, rel, x, values, gagg, gupd = _rows_case(53, bs=11, n_list=17, masked=False)
    point_rows = torch.arange(11)
    bs, n = x.shape[:2]
    sei, set_, sw, _ = rspmm_oracle.sort_edges(ei, et, keep)
    plan = rspmm.Plan(ei, et, n, rel.shape[1], exact_order=False)
    d = lambda t: t.to(dev).contiguous()
    rel_d, x_d, rows_d, gagg_d = d(rel), d(x), d(rows), d(gagg)
    rg, xg = torch.empty_like(rel_d), torch.empty_like(x_d)
    mats = [rspmm.as_mat(t)[1] for t in (rel_d, x_d, rg, xg)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.ultra_rspmm_rows_backward_gather(plan._h, 0, None, ctypes.byref(mats[0]), ctypes.byref(mats[1]), rows_d.data_ptr(), 17,
                                              gagg_d.data_ptr(), None, None, None, ctypes.byref(mats[2]), ctypes.byref(mats[3]), stream)
    assert rc == 0
    for b in (0, 7, 8, 10):       # both chunks of eight samples
        og = torch.zeros(n, 64).index_add_(0, rows[b], gagg[b])
        out = rspmm_oracle.rspmm_forward(sei, set_, sw, rel[b], x[b])
        _, want_rg, want_xg = rspmm_oracle.rspmm_backward(sei, set_, sw, rel[b], x[b], out, og)
        assert (rg[b].cpu() - want_rg).abs().max().item() <= 3e-5 * max(1.0, want_rg.abs().max().item())
        assert (xg[b].cpu() - want_xg).abs().max().item() <= 3e-5 * max(1.0, want_xg.abs().max().item())
    long_rows = torch.randint(0, n, (11, 1025)).to(dev)
    long_g = torch.zeros(11, 1025, 64, device=dev)
    rc = lib.ultra_rspmm_rows_backward_gather(plan._h, 0, None, ctypes.byref(mats[0]), ctypes.byref(mats[1]), long_rows.data_ptr(), 1025,
                                              long_g.data_ptr(), None, None, None, ctypes.byref(mats[2]), ctypes.byref(mats[3]), stream)
    assert rc == ULTRA_ERR_UNSUPPORTED


# ---- the first layer's backward kernel (csrc/onehot_bwd.hip) vs the oracle on the one-hot input it stands for ----
@pytest.mark.parametrize("masked", [False, True])
def test_first_layer_backward_kernel_against_the_oracle(dev, masked):
    """relation_grad and the boundary values' gradient of rspmm._OnehotRSPMM vs rspmm_backward_cpu on the dense one-hot input:
    relation_grad as is; values_grad[b] = input_grad[b, rows[b]] + output_grad[b, rows[b]] (the input IS the boundary that is
    added to the sum, layers.py:199-200)."""
    gen = torch.Generator().manual_seed(61)
    n, e, bs, num_rel = 300, 5000, 4, 6
    ei = torch.randint(0, n, (2, e), generator=gen)
    ei[1, :900] = 5                                       # a hub SOURCE: 900 out-edges of the query node
    et = torch.randint(0, num_rel, (e,), generator=gen)
    rows = torch.tensor([5, 17, 200, 5])
    values = torch.randn(bs, 64, generator=gen)
    rel = torch.randn(bs, num_rel, 64, generator=gen)
    og = torch.randn(bs, n, 64, generator=gen)
    keep = (torch.rand(e, generator=gen) > 0.3).float() if masked else torch.ones(e)
    sei, set_, sw, _ = rspmm_oracle.sort_edges(ei, et, keep)
    want_rg, want_vg = [], []
    for b in range(bs):
        x = torch.zeros(n, 64)
        x[rows[b]] = values[b]
        out = rspmm_oracle.rspmm_forward(sei, set_, sw, rel[b], x)
        _, rg, xg = rspmm_oracle.rspmm_backward(sei, set_, sw, rel[b], x, out, og[b])
        want_rg.append(rg), want_vg.append(xg[rows[b]] + og[b, rows[b]])
    want_rg, want_vg = torch.stack(want_rg), torch.stack(want_vg)
    ei_d, et_d = ei.to(dev), et.to(dev)
    ptr, order, _ = rspmm.out_edge_csr(ei_d, et_d, n)
    got = rspmm._onehot_backward_kernel(ptr, order, ei_d, et_d, keep.to(dev) if masked else None, rel.to(dev), rows.to(dev),
                                        values.to(dev), og.to(dev), True, True)
    assert got is not None
    assert (got[0].cpu() - want_rg).abs().max().item() <= 3e-5 * max(1.0, want_rg.abs().max().item())
    assert (got[1].cpu() - want_vg).abs().max().item() <= 3e-5 * max(1.0, want_vg.abs().max().item())


# ---- the entity model's relation_projection MLPs as one autograd node (csrc/relproj_bwd.hip) ----
@pytest.mark.parametrize("rows_shape,n_layer", [((8, 474), 6), ((3, 37), 2), ((1, 5), 1), ((2, 64), 8)])
def test_relation_projection_node_matches_autograd_of_the_reference_chain(dev, rows_shape, n_layer):
    """dense.RelationProjectionFunction vs torch autograd of nn.Sequential(Linear(64, 64), ReLU, Linear(64, 64)) per layer
    (layers.py:43-47, 80) in fp64 -- outputs, the input gradient and all four parameter gradients of every layer; the fused fp32
    result may not be further from fp64 than a few times torch's own fp32 chain.  One layer's output is left unused (its gradient
    arrives as None)."""
    from ultra_amd import dense
    gen = torch.Generator().manual_seed(n_layer * 100 + rows_shape[1])
    x = torch.randn(*rows_shape, 64, generator=gen)
    params = []
    for _ in range(n_layer):
        params.append([torch.randn(64, 64, generator=gen) / 8, torch.randn(64, generator=gen) / 4,
                       torch.randn(64, 64, generator=gen) / 8, torch.randn(64, generator=gen) / 4])
    gouts = [torch.randn(*rows_shape, 64, generator=gen) for _ in range(n_layer)]
    unused = n_layer - 1 if n_layer > 1 else None

    def chain(dtype, device, fused):
        xs = x.clone().to(device=device, dtype=dtype).requires_grad_()
        ps = [[t.clone().to(device=device, dtype=dtype).requires_grad_() for t in group] for group in params]
        if fused:
            outs = dense.relation_projection_train(xs, [tuple(g) for g in ps])
        else:
            outs = [torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(xs, w0, b0)), w2, b2) for w0, b0, w2, b2 in ps]
        loss = sum((o * g.to(device=device, dtype=dtype)).sum() for l, (o, g) in enumerate(zip(outs, gouts)) if l != unused)
        loss.backward()
        grads = [xs.grad] + [t.grad if t.grad is not None else torch.zeros_like(t) for group in ps for t in group]
        return [o.detach().cpu().double() for o in outs], [g.cpu().double() for g in grads]

    out64, g64 = chain(torch.float64, "cpu", False)
    out32, g32 = chain(torch.float32, dev, False)
    outf, gf = chain(torch.float32, dev, True)
    for a, b in zip(outf, out64):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    for k, (a, r32, r64) in enumerate(zip(gf, g32, g64)):
        scale = max(r64.abs().max().item(), 1e-6)
        err, err_torch = (a - r64).abs().max().item(), (r32 - r64).abs().max().item()
        assert err <= 4 * err_torch + 2e-5 * scale, "gradient %d: |fused - fp64| = %g, |torch fp32 - fp64| = %g (scale %g)" % (
            k, err, err_torch, scale)
    # deterministic
    _, again = chain(torch.float32, dev, True)
    assert all(torch.equal(a, b) for a, b in zip(gf, again))


# ---- the strict sampler kernel (csrc/sampling.hip) against batches recorded from the REFERENCE ----
def test_strict_sampler_kernel_replays_the_reference_batches(dev):
    """tests/golden/negative_sampling.pt: generator state before, and batch returned by, the reference's tasks.negative_sampling
    (strict, a KG with a hub node).  The uniform draws are replayed from that state on the CPU -- torch.rand in the reference's
    order and shapes (tasks.py:57, 65: tails of the first half, then heads of the second) -- and fed to ultra_strict_negatives:
    the kernel returns the reference's ids, id for id."""
    import os
    from ultra_amd import tasks
    from ultra_amd.data import Data
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "negative_sampling.pt"))
    data = Data(edge_index=g["edge_index"], edge_type=g["edge_type"], num_nodes=g["num_nodes"], num_relations=g["num_relations"]).to(dev)
    checked = 0
    for case in g["cases"]:
        if not case["strict"]:
            continue
        batch, n_neg = case["batch"], case["num_negative"]
        half = len(batch) // 2
        torch.set_rng_state(case["rng_state"])
        rand_t = torch.rand(half, n_neg)
        rand_h = torch.rand(len(batch) - half, n_neg)
        pos_h, pos_t, pos_r = batch.to(dev).t()
        neg_t = tasks._strict_negatives_gpu(data, pos_h[:half], pos_r[:half], pos_t[:half], n_neg, known=0, rand=rand_t)
        neg_h = tasks._strict_negatives_gpu(data, pos_t[half:], pos_r[half:], pos_h[half:], n_neg, known=1, rand=rand_h)
        want = case["out"]
        assert torch.equal(neg_t.cpu(), want[:half, 1:, 1]), "tails, %d negatives" % n_neg
        assert torch.equal(neg_h.cpu(), want[half:, 1:, 0]), "heads, %d negatives" % n_neg
        checked += 1
    assert checked == 4


@pytest.mark.parametrize("route", ["hash", "sorted"])
def test_edge_keep_vector_marks_the_edges_the_reference_removes(dev, route):
    """The training step's 0/1 edge vector (ultra_easy_edge_keep: the batch's triples hashed in LDS; ultra_edge_keep_mask: the
    sorted key list) against which edges the REFERENCE's remove_easy_edges left in the graph (tests/golden/easy_edges.pt,
    recorded by gen_golden.py from base_nbfnet.py:54-77)."""
    from tests.test_tasks import easy_edges_golden
    from ultra_amd import dense, models
    data, cases = easy_edges_golden()
    data = data.to(dev)
    models.EASY_EDGE_KEEP_KERNEL = route == "hash"
    try:
        for case in cases:
            model = models.EntityNBFNet(64, [64] * 2, remove_one_hop=case["remove_one_hop"])
            batch = case["batch"].to(dev)
            h, t, r = batch.unbind(-1)                       # (columns of the contiguous batch: stride 3)
            got = model.easy_edge_keep(data, h, t, r)
            assert got.dtype == torch.float32 and torch.equal(got.bool().cpu(), case["keep"])
            got = model.easy_edge_keep(data, h.contiguous(), t.contiguous(), r.contiguous())      # (stride 1)
            assert torch.equal(got.bool().cpu(), case["keep"])
    finally:
        models.EASY_EDGE_KEEP_KERNEL = True
    if route == "hash":
        # the entry itself took the batch (not the fallback), and a list beyond its table is declined, not truncated
        case = cases[0]
        h, t, r = case["batch"].to(dev).unbind(-1)
        assert dense.easy_edge_keep(data.edge_index, data.edge_type, h, t, r, data.num_nodes, data.num_relations) is not None
        big = case["batch"].to(dev).repeat(1, 20, 1)
        h, t, r = big.unbind(-1)
        assert 2 * h.numel() > dense.EDGE_KEEP_MAX_EASY
        assert dense.easy_edge_keep(data.edge_index, data.edge_type, h, t, r, data.num_nodes, data.num_relations) is None


def test_batch_prologue_converts_the_rows_as_the_reference_does(dev):
    """ultra_batch_prologue_rows (h0, r0, the converted rows' candidates, the validity flag) against negative_sample_to_tail as
    restated in the oracle (oracle/ultra_oracle_model.py, base_nbfnet.py:79-86), on the reference-recorded batches of
    tests/golden/negative_sampling.pt (tail rows in the first half, head rows in the second)."""
    from oracle import ultra_oracle_model
    from tests.test_tasks import _negative_sampling_golden
    from ultra_amd import dense
    data, cases = _negative_sampling_golden()
    num_direct = data.num_relations // 2
    for case in cases:
        batch = case["out"]
        h, t, r = ultra_oracle_model.negative_sample_to_tail(*batch.unbind(-1), num_direct)
        pro = dense.batch_prologue(batch.to(dev), num_direct, candidates=True)
        assert torch.equal(pro[1].cpu(), h[:, 0]) and torch.equal(pro[2].cpu(), r[:, 0])
        assert torch.equal(pro.cand.cpu(), t)
        assert bool(pro[4].all())
        assert (h == h[:, :1]).all() and (r == r[:, :1]).all()
    # a row with neither a shared head nor a shared tail is reported, not converted silently
    bad = cases[0]["out"].clone()
    bad[1, 3, 0] += 1
    bad[1, 4, 1] += 1
    pro = dense.batch_prologue(bad.to(dev), num_direct, candidates=True)
    assert pro[4].cpu().tolist() == [1] + [0] + [1] * (len(bad) - 2)


# ---- the training step's readout as one autograd node (csrc/readout_train.hip) ----
@pytest.mark.parametrize("bs,n", [(8, 257), (3, 33), (1, 1), (2, 64), (5, 31)])
def test_readout_node_matches_autograd_of_the_reference_chain(dev, bs, n):
    """dense.ReadoutTrainFunction vs the reference's readout on the candidates' rows -- mlp(cat[hidden, query]) with
    mlp = nn.Sequential(Linear(128, 128), ReLU, Linear(128, 1)) (models.py:120-127, 202-207) -- under torch autograd in fp64: the
    scores (also against the oracle's fp32 restatement of torch's Linear, oracle/torch_math_oracle.py), the gradients of the hidden
    rows, the query and the four parameters.  The fused fp32 result may not be further from fp64 than a few times torch's own fp32
    chain; two runs give the same bits."""
    from oracle import torch_math_oracle
    from ultra_amd import dense
    gen = torch.Generator().manual_seed(bs * 1000 + n)
    hid = torch.randn(bs, n, 64, generator=gen)
    query = torch.randn(bs, 64, generator=gen)
    params = [torch.randn(128, 128, generator=gen) / 11, torch.randn(128, generator=gen) / 4,
              torch.randn(1, 128, generator=gen) / 11, torch.randn(1, generator=gen)]
    gout = torch.randn(bs, n, generator=gen)

    def chain(dtype, device, fused):
        leaves = [t.clone().to(device=device, dtype=dtype).requires_grad_() for t in (hid, query, *params)]
        h, q, w1, b1, w2, b2 = leaves
        if fused:
            score = dense.ReadoutTrainFunction.apply(h, q, w1, b1, w2, b2)
        else:
            feature = torch.cat([h, q.unsqueeze(1).expand(-1, n, -1)], dim=-1)
            score = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(feature, w1, b1)), w2, b2).squeeze(-1)
        (score * gout.to(device=device, dtype=dtype)).sum().backward()
        return score.detach().cpu().double(), [t.grad.cpu().double() for t in leaves]

    s64, g64 = chain(torch.float64, "cpu", False)
    s32, g32 = chain(torch.float32, dev, False)
    sf, gf = chain(torch.float32, dev, True)
    assert sf.shape == (bs, n)
    assert (sf - s64).abs().max().item() <= 2e-5 * max(1.0, s64.abs().max().item())
    feature = torch.cat([hid, query.unsqueeze(1).expand(-1, n, -1)], dim=-1).reshape(-1, 128)
    want = torch_math_oracle.linear(torch.relu(torch_math_oracle.linear(feature, params[0], params[1])), params[2], params[3])
    assert (sf.float() - want.view(bs, n)).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    for k, (a, r32, r64) in enumerate(zip(gf, g32, g64)):
        assert a.shape == r64.shape
        scale = max(r64.abs().max().item(), 1e-6)
        err, err_torch = (a - r64).abs().max().item(), (r32 - r64).abs().max().item()
        assert err <= 4 * err_torch + 2e-5 * scale, "gradient %d: |fused - fp64| = %g, |torch fp32 - fp64| = %g (scale %g)" % (
            k, err, err_torch, scale)
    s_again, g_again = chain(torch.float32, dev, True)
    assert torch.equal(sf, s_again) and all(torch.equal(a, b) for a, b in zip(gf, g_again))


def test_training_forward_takes_the_readout_node(dev):
    """EntityNBFNet.forward under autograd with the ULTRA readout goes through ReadoutTrainFunction; the switch selects torch's chain,
    same scores to rounding."""
    from ultra_amd import dense, models, synthetic
    data = synthetic.make_kg(num_node=300, num_triple=3000, num_relation_base=5, num_test=16, seed=5).to(dev)
    torch.manual_seed(0)
    model = models.Ultra(**synthetic.default_model_cfg()).to(dev).train()
    batch = torch.stack([data.edge_index[0, :4], data.edge_index[1, :4], data.edge_type[:4]], dim=-1)
    from ultra_amd import tasks
    batch = tasks.negative_sampling(data, batch, 16, strict=True)
    score = model(data, batch)
    node = score.grad_fn
    for _ in range(3):                      # (the score comes back through .view(shape))
        if type(node).__name__ == "ReadoutTrainFunctionBackward":
            break
        node = node.next_functions[0][0]
    assert type(node).__name__ == "ReadoutTrainFunctionBackward"
    dense.READOUT_TRAIN_NODE = False
    try:
        other = model(data, batch)
    finally:
        dense.READOUT_TRAIN_NODE = True
    assert torch.allclose(score, other, rtol=1e-5, atol=1e-5)
