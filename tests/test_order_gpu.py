"""Reference-order plans (ULTRA_PLAN_EXACT_ORDER) on the order kernels (csrc/rspmm_order_kernels.hpp): every
aggregate -- sums included -- must equal the oracle's sequential loop (rspmm.cpp:50-75) BIT FOR BIT, for group
items and for chain rows (whole-workgroup walk through the LDS ring) alike."""
import itertools

import pytest
import torch

from oracle import rspmm_oracle
from tests import helpers

pytestmark = pytest.mark.gpu

SUMS = ["add", "min", "max"]
MULS = ["mul", "add"]
CASES = [
    dict(num_node=50, num_edge=400, num_relation=5, seed=0),
    dict(num_node=64, num_edge=300, num_relation=3, seed=1, hub=(7, 700)),          # one chain row at the default threshold
    dict(num_node=40, num_edge=100, num_relation=4, seed=2, empty_rows=10),
    dict(num_node=30, num_edge=200, num_relation=1, seed=3, duplicates=50),
    dict(num_node=5, num_edge=0, num_relation=2, seed=4),
    dict(num_node=1, num_edge=17, num_relation=2, seed=5),
    dict(num_node=700, num_edge=9000, num_relation=800, seed=6, hub=(3, 1500)),     # relation slice does not fit LDS
    dict(num_node=100, num_edge=20000, num_relation=4, seed=7),                    # dense: rows of ~200 edges
    dict(num_node=300, num_edge=2000, num_relation=9, seed=8, hub=(11, 4321)),      # 73 chunks, partial last chunk
]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _reset():
    from ultra_amd import rspmm
    rspmm.set_tuning()
    rspmm.set_plan_defaults()
    yield
    rspmm.set_tuning()
    rspmm.set_plan_defaults()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("sum,mul", list(itertools.product(SUMS, MULS)))
@pytest.mark.parametrize("dtype,dim", [(torch.float32, 64), (torch.float32, 192), (torch.float64, 128)])
def test_bit_exact_against_oracle(dev, case, sum, mul, dtype, dim):
    from ultra_amd.rspmm import Plan
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    rel, x, w = helpers.features(N, R, dim, E, dtype=dtype, seed=case["seed"])
    plan = Plan(ei, et, N, R, exact_order=True)
    want = rspmm_oracle.generalized_rspmm(ei, et, w, rel, x, sum=sum, mul=mul)
    got = plan.forward(rel.to(dev), x.to(dev), edge_weight=w.to(dev), sum=sum, mul=mul).cpu()
    assert torch.equal(got, want), "max |d| = %g" % (got - want).abs().max().item()
    ones = torch.ones(E, dtype=dtype)
    want1 = rspmm_oracle.generalized_rspmm(ei, et, ones, rel, x, sum=sum, mul=mul)
    got1 = plan.forward(rel.to(dev), x.to(dev), edge_weight=None, sum=sum, mul=mul).cpu()
    assert torch.equal(got1, want1)


@pytest.mark.parametrize("case", [CASES[1], CASES[7], CASES[8]])
@pytest.mark.parametrize("chain_min", [4, 59, 60, 61, 1000000])
def test_chain_threshold_does_not_change_a_bit(dev, case, chain_min):
    """Whether a row is walked by one lane group or by the workgroup's chain pipeline is a scheduling decision."""
    from ultra_amd.rspmm import Plan
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    rel, x, w = helpers.features(N, R, 128, E, seed=case["seed"])
    want = rspmm_oracle.generalized_rspmm(ei, et, w, rel, x, sum="add", mul="mul")
    plan = Plan(ei, et, N, R, exact_order=True, seg_len=chain_min)
    info = plan.info()
    got = plan.forward(rel.to(dev), x.to(dev), edge_weight=w.to(dev)).cpu()
    assert torch.equal(got, want), info
    deg = torch.bincount(ei[0], minlength=N)
    assert info["n_chain_row"] == int((deg > chain_min).sum())


@pytest.mark.parametrize("sum", SUMS)
@pytest.mark.parametrize("layout", ["node_major", "batch_major", "shared_relation"])
@pytest.mark.parametrize("boundary", ["none", "tensor", "point"])
def test_layouts_and_boundaries(dev, sum, layout, boundary):
    from ultra_amd.rspmm import Plan
    case = CASES[8]
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    bs, d = 3, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(bs, N, d, generator=g)
    rel = torch.randn(1 if layout == "shared_relation" else bs, R, d, generator=g).expand(bs, -1, -1)
    bnd = torch.randn(bs, N, d, generator=g)
    rows = torch.tensor([11, 0, N - 1])
    vals = torch.randn(bs, d, generator=g)
    if boundary == "point":
        bnd = torch.zeros(bs, N, d)
        bnd[torch.arange(bs), rows] = vals
    ones = torch.ones(E)
    # the reference's data flow: node-major (N, batch * d), boundary combined after the aggregate (layers.py:190-207)
    xn, reln, bndn = (t.transpose(0, 1).flatten(1).contiguous() for t in (x, rel, bnd))
    want = rspmm_oracle.generalized_rspmm(ei, et, ones, reln, xn, sum=sum, mul="mul")
    if boundary != "none":
        want = want + bndn if sum == "add" else (torch.max(want, bndn) if sum == "max" else torch.min(want, bndn))
    plan = Plan(ei, et, N, R, exact_order=True)
    kw = {}
    if boundary == "tensor":
        kw["boundary"] = bndn.to(dev) if layout == "node_major" else bnd.to(dev)
    if layout == "node_major":
        if boundary == "point":
            pytest.skip("point boundaries ride on the batch-major layout")
        got = plan.forward(reln.to(dev), xn.to(dev), sum=sum, **kw).cpu()
    else:
        if boundary == "point":
            kw["point"] = (rows.to(dev), vals.to(dev))
        rel_dev = rel[:1].to(dev).expand(bs, -1, -1) if layout == "shared_relation" else rel.contiguous().to(dev)
        got = plan.forward(rel_dev, x.to(dev), sum=sum, **kw).cpu().transpose(0, 1).flatten(1)
    assert torch.equal(got, want)


@pytest.mark.parametrize("sum", ["min", "max"])
@pytest.mark.parametrize("walk", ["streams", "units", "weighted", "f64"])
@pytest.mark.parametrize("case", [CASES[2], CASES[8]])
def test_point_boundary_under_min_max_meets_zero_off_the_query_rows(dev, sum, walk, case):
    """layers.py:206-207: max(update, boundary) against a tensor that is zero off the query rows -- the point form must give
    every other row max(update, 0) (an edge-less row: 0, not -FLT_MAX), on the assembly walk, the C++ unit walk, the
    weighted kernels and the fp64 kernels, chain rows included."""
    from ultra_amd import rspmm
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    dtype = torch.float64 if walk == "f64" else torch.float32
    bs, d = 3, 64
    g = torch.Generator().manual_seed(6)
    x = torch.randn(bs, N, d, generator=g, dtype=torch.float64).to(dtype)
    rel = torch.randn(bs, R, d, generator=g, dtype=torch.float64).to(dtype)
    w = (torch.rand(E, generator=g, dtype=torch.float64) + 0.5).to(dtype) if walk == "weighted" else torch.ones(E, dtype=dtype)
    rows = torch.tensor([11 % N, 0, N - 1])      # (N - 1: an edge-less row in CASES[2])
    vals = torch.randn(bs, d, generator=g, dtype=torch.float64).to(dtype)
    bnd = torch.zeros(bs, N, d, dtype=dtype)
    bnd[torch.arange(bs), rows] = vals
    xn, reln, bndn = (t.transpose(0, 1).flatten(1).contiguous() for t in (x, rel, bnd))
    want = rspmm_oracle.generalized_rspmm(ei, et, w, reln, xn, sum=sum, mul="mul")
    want = torch.max(want, bndn) if sum == "max" else torch.min(want, bndn)
    plan = rspmm.Plan(ei, et, N, R, exact_order=True)
    if walk == "units":
        rspmm.set_tuning(unit_walk=1)
    got = plan.forward(rel.to(dev), x.to(dev), edge_weight=w.to(dev) if walk == "weighted" else None, sum=sum,
                       point=(rows.to(dev), vals.to(dev)))
    assert got is not None
    assert torch.equal(got.cpu().transpose(0, 1).flatten(1), want)
    # a plan that re-associates declines the point form under min / max (the caller passes the tensor)
    loose = rspmm.Plan(ei, et, N, R, exact_order=False)
    assert loose.forward(rel.to(dev), x.to(dev), sum=sum, point=(rows.to(dev), vals.to(dev))) is None


@pytest.mark.parametrize("sum", ["max", "min"])
@pytest.mark.parametrize("miss", ["general_walk", "row_len_30"])
def test_point_boundary_under_min_max_is_declined_off_the_order_kernels(dev, sum, miss):
    """a reference-order plan whose call misses the order kernels (general_walk tuning, a row length that is no
    multiple of four elements) must not run the general walk with a point boundary under min / max -- that kernel has no fill for
    the rows off the boundary row.  The call answers None (ULTRA_ERR_UNSUPPORTED) and the layer's route -- the boundary as a
    tensor -- gives the reference's values."""
    from ultra_amd import rspmm
    case = CASES[2]
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    bs, d = 2, (30 if miss == "row_len_30" else 64)
    g = torch.Generator().manual_seed(16)
    x, rel = torch.randn(bs, N, d, generator=g), torch.randn(bs, R, d, generator=g)
    rows, vals = torch.tensor([3 % N, N - 1]), torch.randn(bs, d, generator=g)
    bnd = torch.zeros(bs, N, d)
    bnd[torch.arange(bs), rows] = vals
    xn, reln, bndn = (t.transpose(0, 1).flatten(1).contiguous() for t in (x, rel, bnd))
    want = rspmm_oracle.generalized_rspmm(ei, et, torch.ones(E), reln, xn, sum=sum, mul="mul")
    want = torch.max(want, bndn) if sum == "max" else torch.min(want, bndn)
    plan = rspmm.Plan(ei, et, N, R, exact_order=True)
    if miss == "general_walk":
        rspmm.set_tuning(general_walk=1)
    assert plan.forward(rel.to(dev), x.to(dev), sum=sum, point=(rows.to(dev), vals.to(dev))) is None
    got = plan.forward(rel.to(dev), x.to(dev), sum=sum, boundary=bnd.to(dev))
    assert torch.allclose(got.cpu().transpose(0, 1).flatten(1), want, rtol=1e-6, atol=1e-6)
    # the edge-less row N - 1 meets the tensor's zero (sample 0) or its value (sample 1), never -+FLT_MAX
    assert got[0, N - 1].abs().max().item() == 0.0


@pytest.mark.parametrize("grid", [1, 3, 8, 77, 256, 1000])
def test_any_grid_same_bits(dev, grid):
    """Schedules are built per workgroups-per-span; more spans than workgroups loop inside the workgroup."""
    from ultra_amd import rspmm
    case = CASES[8]
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    rel, x, w = helpers.features(N, R, 64 * 5, E, seed=1)      # 5 spans
    want = rspmm_oracle.generalized_rspmm(ei, et, w, rel, x)
    plan = rspmm.Plan(ei, et, N, R, exact_order=True)
    rspmm.set_tuning(grid=grid)
    got = plan.forward(rel.to(dev), x.to(dev), edge_weight=w.to(dev)).cpu()
    assert torch.equal(got, want)
    rspmm.set_tuning(grid=grid, rel_lds=0)
    got = plan.forward(rel.to(dev), x.to(dev), edge_weight=w.to(dev)).cpu()
    assert torch.equal(got, want)


def test_two_streams_share_one_plan(dev):
    """The plan is immutable after upload (no partial slots, no weight scratch): concurrent launches on two streams
    with different operands do not disturb one another."""
    from ultra_amd.rspmm import Plan
    case = CASES[8]
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    plan = Plan(ei, et, N, R, exact_order=True)
    ops = []
    for seed in (1, 2):
        rel, x, w = helpers.features(N, R, 256, E, seed=seed)
        ops.append((rel, x, w, rspmm_oracle.generalized_rspmm(ei, et, w, rel, x)))
    dev_ops = [(r.to(dev), x.to(dev), w.to(dev)) for r, x, w, _ in ops]
    plan.forward(dev_ops[0][0], dev_ops[0][1], edge_weight=dev_ops[0][2])      # upload + schedule
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for it in range(20):
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[k].append(plan.forward(dev_ops[k][0], dev_ops[k][1], edge_weight=dev_ops[k][2]))
    torch.cuda.synchronize()
    for k in range(2):
        for o in outs[k]:
            assert torch.equal(o.cpu(), ops[k][3])


def test_headline_shape_bit_exact(dev):
    """FB15k237-shaped entity graph, batch 8 (the benchmark's rspmm call): 97 chain rows up to 9,067 edges."""
    from ultra_amd import synthetic
    from ultra_amd.rspmm import Plan
    data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234, relation_graph=False)
    N, R, E = data.num_nodes, data.num_relations, data.num_edges
    g = torch.Generator().manual_seed(0)
    bs = 8
    x = torch.randn(bs, N, 64, generator=g)
    rel = torch.randn(bs, R, 64, generator=g)
    want = rspmm_oracle.generalized_rspmm(data.edge_index, data.edge_type, torch.ones(E), rel.transpose(0, 1).flatten(1),
                                          x.transpose(0, 1).flatten(1))
    plan = Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
    got = plan.forward(rel.to(dev), x.to(dev)).cpu().transpose(0, 1).flatten(1)
    assert torch.equal(got, want)


def _relation_like_graph(num_node, fill, seed, num_type=4):
    """Edges listed type block after type block (hh, tt, ht, th in tasks.py:186-189), each (row, col, type) at most once."""
    g = torch.Generator().manual_seed(seed)
    blocks = []
    for t in range(num_type):
        mask = torch.rand(num_node, num_node, generator=g) < fill
        rc = mask.nonzero().t()
        blocks.append(torch.cat([rc, torch.full((1, rc.shape[1]), t)]))
    e = torch.cat(blocks, dim=1)
    return e[:2].contiguous(), e[2].contiguous()


@pytest.mark.parametrize("num_node,fill", [(100, 0.9), (474, 0.995), (37, 0.5), (130, 1.0)])
@pytest.mark.parametrize("boundary", ["none", "tensor", "point"])
def test_dense_layer_aggregate_in_reference_order_is_bit_exact(dev, num_node, fill, boundary):
    """ultra_nbf_dense_layer with ULTRA_LAYER_REFERENCE_ORDER: with W = [0 | I], no bias / LayerNorm / ReLU the layer's
    output IS its aggregate, which must equal the oracle's rspmm (+ boundary) bit for bit."""
    from torch import nn
    from ultra_amd.rspmm import Plan
    ei, et = _relation_like_graph(num_node, fill, seed=num_node)
    N, E, bs = num_node, ei.shape[1], 3
    g = torch.Generator().manual_seed(1)
    x = torch.randn(bs, N, 64, generator=g)
    rel = torch.randn(bs, 4, 64, generator=g)
    bnd = torch.randn(bs, N, 64, generator=g)
    rows = torch.tensor([5, 0, N - 1])
    vals = torch.randn(bs, 64, generator=g)
    if boundary == "point":
        bnd = torch.zeros(bs, N, 64)
        bnd[torch.arange(bs), rows] = vals
    want = rspmm_oracle.generalized_rspmm(ei, et, torch.ones(E), rel.transpose(0, 1).flatten(1), x.transpose(0, 1).flatten(1))
    if boundary != "none":
        want = want + bnd.transpose(0, 1).flatten(1)
    want = want.view(N, bs, 64).transpose(0, 1)
    plan = Plan(ei, et, N, 4, exact_order=True)
    assert plan.dense is not None and plan.dense.info()["dense_order_bytes"] > 0
    lin = nn.Linear(128, 64, bias=False)
    with torch.no_grad():
        lin.weight.zero_()
        lin.weight[:, 64:] = torch.eye(64)
    lin = lin.to(dev)
    kw = {}
    if boundary == "tensor":
        kw["boundary"] = bnd.to(dev)
    elif boundary == "point":
        kw["point"] = (rows.to(dev), vals.to(dev))
    got = plan.fused_layer(rel.to(dev), x.to(dev), lin, layer_norm=None, relu=False, residual=False, **kw)
    assert got is not None
    assert torch.equal(got.cpu(), want), "max |d| = %g" % (got.cpu() - want).abs().max().item()
    # the order kernels agree (they serve graphs that do not qualify for the dense format)
    sparse = Plan(ei, et, N, 4, exact_order=True, dense=False)
    got2 = sparse.forward(rel.to(dev), x.to(dev), **kw).cpu()
    assert torch.equal(got2, want)


def test_dense_layer_declines_graphs_it_cannot_order(dev):
    """Parallel edges not sorted by type (or repeated): no reference-order dense twin, the order kernels serve the plan."""
    from ultra_amd.rspmm import Plan
    ei, et = _relation_like_graph(60, 0.9, seed=3)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))
    shuffled = Plan(ei[:, perm], et[perm], 60, 4, exact_order=True)
    assert shuffled.dense is None
    dup = Plan(torch.cat([ei, ei[:, :5]], dim=1), torch.cat([et, et[:5]]), 60, 4, exact_order=True)
    assert dup.dense is None
    assert Plan(ei, et, 60, 4, exact_order=True).dense is not None


@pytest.mark.parametrize("residual,layer_norm,relu", [(True, True, True), (False, True, False), (True, False, True)])
def test_dense_order_layer_matches_rspmm_plus_update(dev, residual, layer_norm, relu):
    """The whole layer against order-kernel aggregate + the stand-alone update kernel (different product blocking: 1e-5)."""
    from torch import nn
    from ultra_amd import dense
    from ultra_amd.rspmm import Plan
    ei, et = _relation_like_graph(474, 0.995, seed=9)
    N, bs = 474, 8
    g = torch.Generator().manual_seed(2)
    x = torch.randn(bs, N, 64, generator=g).to(dev)
    rel = torch.randn(bs, 4, 64, generator=g).to(dev) * 0.1
    point = (torch.arange(bs).to(dev) * 3, torch.randn(bs, 64, generator=g).to(dev))
    torch.manual_seed(0)
    lin = nn.Linear(128, 64).to(dev)
    ln = nn.LayerNorm(64).to(dev) if layer_norm else None
    plan = Plan(ei, et, N, 4, exact_order=True)
    got = plan.fused_layer(rel, x, lin, layer_norm=ln, relu=relu, residual=residual, point=point)
    agg = Plan(ei, et, N, 4, exact_order=True, dense=False).forward(rel, x, point=point)
    want = lin(torch.cat([x, agg], dim=-1))
    if ln is not None:
        want = ln(want)
    if relu:
        want = torch.relu(want)
    if residual:
        want = want + x
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5 * max(scale, 1.0)


@pytest.mark.parametrize("N", [474, 40, 17, 33, 100])
def test_dense_order_layer_with_two_row_tiles_per_workgroup_keeps_its_bits(dev, N, monkeypatch):
    """The relation-graph layer with TWO 16-row tiles per workgroup over one stream of B operands (dense_order_layer_kernel<2>,
    ULTRA_DOL_TILES=2 -- VERDICT r4 item 3a: built, measured slower in the step, not the default): the same bits as one tile per
    workgroup -- an even and an odd number of tiles, a last tile with fewer than 16 rows."""
    from torch import nn
    from ultra_amd.rspmm import Plan
    ei, et = _relation_like_graph(N, 0.97, seed=N)
    bs = 5
    g = torch.Generator().manual_seed(N)
    x = torch.randn(bs, N, 64, generator=g).to(dev)
    rel = torch.randn(bs, 4, 64, generator=g).to(dev) * 0.1
    point = (torch.arange(bs).to(dev) * 3 % N, torch.randn(bs, 64, generator=g).to(dev))
    tensor_bnd = torch.randn(bs, N, 64, generator=g).to(dev)
    torch.manual_seed(1)
    lin, ln = nn.Linear(128, 64).to(dev), nn.LayerNorm(64).to(dev)
    plan = Plan(ei, et, N, 4, exact_order=True)
    assert plan.dense is not None
    outs = {}
    for tiles in ("1", "2"):
        monkeypatch.setenv("ULTRA_DOL_TILES", tiles)
        outs[tiles] = (plan.fused_layer(rel, x, lin, layer_norm=ln, relu=True, residual=True, point=point),
                       plan.fused_layer(rel, x, lin, layer_norm=None, relu=False, residual=False, boundary=tensor_bnd))
    assert all(o is not None for pair in outs.values() for o in pair)
    assert torch.equal(outs["1"][0], outs["2"][0]) and torch.equal(outs["1"][1], outs["2"][1])
    # the four-workgroups-a-CU form (round 6: <= 128 registers, late operands of phases 2 / 3; the default where the launch shares
    # the chip): the same bits again
    # ... and as ONE 1024-thread workgroup of four such quartets (the default there: packs the chains onto few CUs) -- graphs whose
    # tile count is not a multiple of four leave quartets without a tile
    monkeypatch.setenv("ULTRA_DOL_TILES", "1")
    for form in ("1", "2"):
        monkeypatch.setenv("ULTRA_DOL_LEAN", form)
        lean = (plan.fused_layer(rel, x, lin, layer_norm=ln, relu=True, residual=True, point=point),
                plan.fused_layer(rel, x, lin, layer_norm=None, relu=False, residual=False, boundary=tensor_bnd))
        assert torch.equal(lean[0], outs["1"][0]) and torch.equal(lean[1], outs["1"][1]), form
    monkeypatch.setenv("ULTRA_DOL_LEAN", "0")


@pytest.mark.parametrize("sum,mul", list(itertools.product(SUMS, MULS)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("surface", ["ctypes", "pybind"])
def test_all_twelve_reference_exports(dev, sum, mul, dtype, surface):
    """rspmm_<sum>_<mul>_{forward,backward}_cuda (rspmm.h:63-105) -- through the ctypes namespace ultra_amd.rspmm.rspmm and
    through the pybind11 module `rspmm` -- against the oracle: forward bit for bit (the stateless entries build
    reference-order plans), gradients to 1e-4 relative (rspmm.cpp:77-119 accumulates under mutexes: no order to match)."""
    from ultra_amd import build, rspmm as R_
    ns = R_.rspmm if surface == "ctypes" else build.load_torch_binding()
    case = CASES[1]
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    rel, x, w = helpers.features(N, R, 64, E, dtype=dtype, seed=3)
    sei, set_, sw, _ = rspmm_oracle.sort_edges(ei, et, w)
    d = lambda t: t.to(dev)
    want = rspmm_oracle.rspmm_forward(sei, set_, sw, rel, x, sum=sum, mul=mul)
    got = getattr(ns, "rspmm_%s_%s_forward_cuda" % (sum, mul))(d(sei), d(set_), d(sw), d(rel), d(x)).cpu()
    assert torch.equal(got, want)
    g = torch.Generator().manual_seed(9)
    og = torch.randn(want.shape, generator=g, dtype=torch.float64).to(dtype)
    wg, rg, xg = rspmm_oracle.rspmm_backward(sei, set_, sw, rel, x, want, og, sum=sum, mul=mul)
    gwg, grg, gxg = getattr(ns, "rspmm_%s_%s_backward_cuda" % (sum, mul))(d(sei), d(set_), d(sw), d(rel), d(x), d(want), d(og))
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(gxg.cpu(), xg, **tol)
    torch.testing.assert_close(grg.cpu(), rg, **tol)
    torch.testing.assert_close(gwg.cpu(), wg, **tol)
    if surface == "pybind":
        with pytest.raises(RuntimeError, match="Expect sorted"):
            ns.rspmm_add_mul_forward_cuda(d(ei), d(et), d(w), d(rel), d(x))
