"""The fine-tuning path (BASELINE.json config 5) on the GPU: the fused layer-update backward against torch autograd of the
plain op chain (layers.py:233-240), edge dropout as a keep mask against the reference's filtered graph
(base_nbfnet.py:54-77), and a whole training step."""
import pytest
import torch
from torch.nn import functional as F

from tests.test_oracle_model import load_golden
from ultra_amd import dense, layers, models, rspmm, synthetic, tasks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def plain_update(x, agg, w, b, g, beta, eps, ln, relu, residual):
    out = F.linear(torch.cat([x, agg], dim=-1), w, b)
    if ln:
        out = F.layer_norm(out, (64,), g, beta, eps)
    if relu:
        out = F.relu(out)
    return out + x if residual else out


@pytest.mark.parametrize("rows", [1, 5, 32, 1000, 14541 * 8])
@pytest.mark.parametrize("ln,relu,residual", [(True, True, True), (True, True, False), (False, True, True), (True, False, False),
                                               (False, False, False)])
def test_conv_update_backward_matches_autograd(dev, rows, ln, relu, residual):
    """Gradients of the fused node vs autograd of the plain chain in fp64 (what both fp32 paths approximate); the fused
    fp32 gradient may not be further from it than a few times torch's own fp32 chain."""
    if rows > 1000 and not (ln and relu and residual):
        pytest.skip("the large shape runs the ULTRA configuration only")
    gen = torch.Generator().manual_seed(rows * 8 + ln * 4 + relu * 2 + residual)
    x = torch.randn(rows, 64, generator=gen)
    agg = torch.randn(rows, 64, generator=gen) * 3
    w = torch.randn(64, 128, generator=gen) / 11
    b = torch.randn(64, generator=gen) / 10
    g = 1 + torch.randn(64, generator=gen) / 10
    beta = torch.randn(64, generator=gen) / 10
    gout = torch.randn(rows, 64, generator=gen)
    eps = 1e-5

    def run(dtype, device, fused):
        leaves = [t.clone().to(device=device, dtype=dtype).requires_grad_() for t in (x, agg, w, b, g, beta)]
        if fused:
            flags = (1 if ln else 0) | (2 if relu else 0) | (4 if residual else 0)
            out = dense.ConvUpdateFunction.apply(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4] if ln else None,
                                                 leaves[5] if ln else None, eps, flags)
        else:
            out = plain_update(*leaves, eps, ln, relu, residual)
        out.backward(gout.to(device=device, dtype=dtype))
        return out.detach().cpu().double(), [t.grad.cpu().double() if t.grad is not None else None for t in leaves]

    out64, g64 = run(torch.float64, "cpu", False)
    out32, g32 = run(torch.float32, dev, False)
    outf, gf = run(torch.float32, dev, True)
    assert (outf - out64).abs().max().item() <= 2e-5 * max(1.0, out64.abs().max().item())
    for name, a, r32, r64 in zip(("x", "agg", "weight", "bias", "ln_weight", "ln_bias"), gf, g32, g64):
        if r64 is None or (not ln and name.startswith("ln_")):
            assert a is None or not ln
            continue
        scale = max(r64.abs().max().item(), 1e-6)
        err = (a - r64).abs().max().item()
        err_torch = (r32 - r64).abs().max().item()
        assert err <= 4 * err_torch + 2e-5 * scale, "%s: |fused - fp64| = %g, |torch fp32 - fp64| = %g (scale %g)" % (
            name, err, err_torch, scale)


def test_conv_update_backward_is_deterministic(dev):
    gen = torch.Generator().manual_seed(5)
    x, agg, gout = (torch.randn(50000, 64, generator=gen).to(dev) for _ in range(3))
    w = (torch.randn(64, 128, generator=gen) / 11).to(dev)
    vec = [torch.randn(64, generator=gen).to(dev) for _ in range(3)]
    grads = []
    for _ in range(2):
        leaves = [t.clone().requires_grad_() for t in (x, agg, w, *vec)]
        dense.ConvUpdateFunction.apply(*leaves, 1e-5, 7).backward(gout)
        grads.append([t.grad.clone() for t in leaves])
    for a, b in zip(*grads):
        assert torch.equal(a, b)         # partial sums are combined in a fixed order: no atomics


def test_layer_update_uses_the_fused_node_under_grad(dev):
    layer = layers.GeneralizedRelationalConv(64, 64, 10, 64, "distmult", "sum", layer_norm=True, activation="relu",
                                             project_relations=True).to(dev)
    x = torch.randn(3, 77, 64, device=dev, requires_grad=True)
    agg = torch.randn(3, 77, 64, device=dev, requires_grad=True)
    out = layer.update(agg, x, residual=True)
    assert type(out.grad_fn).__name__ == "ConvUpdateFunctionBackward"
    want = plain_update(x, agg, layer.linear.weight, layer.linear.bias, layer.layer_norm.weight, layer.layer_norm.bias,
                        layer.layer_norm.eps, True, True, True)
    assert torch.allclose(out, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("one_hop", [False, True])
def test_edge_keep_mask_matches_edge_match(dev, one_hop):
    data = synthetic.make_kg(num_node=500, num_triple=6000, num_relation_base=7, num_test=16, seed=11).to(dev)
    model = models.EntityNBFNet(64, [64] * 2, remove_one_hop=one_hop)
    batch = torch.stack([data.edge_index[0, :6], data.edge_index[1, :6], data.edge_type[:6]], dim=-1)
    torch.manual_seed(1)
    neg = tasks.negative_sampling(data, batch, 300, strict=True)         # 6 * 2 * 301 = 3612 easy edges
    h, t, r = neg.unbind(-1)
    want = model.easy_edge_mask(data, h, t, r)
    got = model.easy_edge_keep(data, h, t, r)
    assert got.dtype == torch.float32 and torch.equal(got.bool(), want)
    assert 0 < int((~want).sum()) < data.num_edges
    # beyond the kernel's key table: the torch path serves
    big = neg.repeat(1, 6, 1)
    h, t, r = big.unbind(-1)
    assert h.numel() * 2 > dense.EDGE_KEEP_MAX_EASY
    assert torch.equal(model.easy_edge_keep(data, h, t, r).bool(), want)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("sum", ["add", "min", "max"])
@pytest.mark.parametrize("mul", ["mul", "add"])
def test_masked_forward_equals_the_filtered_graph(dev, exact, sum, mul):
    """An edge with keep == 0 is ABSENT: same output as the reference's filtered copy of the graph (bit for bit with
    reference-order plans and for min / max, where the order of the reduction does not matter)."""
    gen = torch.Generator().manual_seed(3)
    n, e, r, bs = 400, 9000, 6, 3
    ei = torch.randint(0, n, (2, e), generator=gen)
    ei[0, :700] = 7                                        # a hub row (chain path of the reference-order kernels)
    et = torch.randint(0, r, (e,), generator=gen)
    keep = torch.rand(e, generator=gen) > 0.3
    keep[ei[0] == 11] = False                              # a row that loses every edge
    rel = torch.randn(bs, r, 64, generator=gen).to(dev)
    x = torch.randn(bs, n, 64, generator=gen).relu().to(dev)          # exact zeros: 0-weight and absent differ under max
    full = rspmm.Plan(ei, et, n, r, exact_order=exact)
    part = rspmm.Plan(ei[:, keep], et[keep], n, r, exact_order=exact)
    got = full.forward(rel, x, edge_weight=keep.float().to(dev), sum=sum, mul=mul, keep=True)
    want = part.forward(rel, x, sum=sum, mul=mul)
    if exact or sum != "add":
        assert torch.equal(got, want)
    else:
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-4)
    if sum != "add":
        weighted = full.forward(rel, x, edge_weight=keep.float().to(dev), sum=sum, mul=mul)      # zero WEIGHTS: value 0 enters
        assert not torch.equal(weighted, want)


@pytest.mark.parametrize("sum", ["add", "max"])
def test_masked_backward_equals_the_filtered_graph(dev, sum):
    gen = torch.Generator().manual_seed(4)
    n, e, r, bs = 300, 5000, 5, 2
    ei = torch.randint(0, n, (2, e), generator=gen)
    et = torch.randint(0, r, (e,), generator=gen)
    keep = torch.rand(e, generator=gen) > 0.25
    rel0 = torch.randn(bs, r, 64, generator=gen)
    x0 = torch.randn(bs, n, 64, generator=gen)
    gout = torch.randn(bs, n, 64, generator=gen).to(dev)
    full = rspmm.Plan(ei, et, n, r)
    part = rspmm.Plan(ei[:, keep], et[keep], n, r)

    def grads(plan, w, keep_flag):
        rel, x = rel0.clone().to(dev).requires_grad_(), x0.clone().to(dev).requires_grad_()
        rspmm.plan_rspmm(plan, rel, x, w, sum=sum, mul="mul", keep=keep_flag).backward(gout)
        return rel.grad, x.grad

    got = grads(full, keep.float().to(dev), True)
    want = grads(part, None, False)
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * b.abs().max().item())


def test_fused_boundary_gradient(dev):
    gen = torch.Generator().manual_seed(6)
    n, e, r, bs = 200, 3000, 4, 2
    ei = torch.randint(0, n, (2, e), generator=gen)
    et = torch.randint(0, r, (e,), generator=gen)
    plan = rspmm.Plan(ei, et, n, r)
    leaves = [t.to(dev).requires_grad_() for t in (torch.randn(bs, r, 64, generator=gen), torch.randn(bs, n, 64, generator=gen),
                                                    torch.randn(bs, n, 64, generator=gen))]
    gout = torch.randn(bs, n, 64, generator=gen).to(dev)
    rspmm.plan_rspmm(plan, leaves[0], leaves[1], boundary=leaves[2]).backward(gout)
    fused = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    (rspmm.plan_rspmm(plan, leaves[0], leaves[1]) + leaves[2]).backward(gout)
    for a, t in zip(fused, leaves):
        assert torch.equal(a, t.grad)


@pytest.mark.parametrize("aggr", ["sum", "max", "mean", "pna"])
def test_training_forward_equals_the_filtered_graph(dev, aggr):
    """model.train(): the batch's own edges are dropped through the keep mask; same scores as the reference's route
    (remove_easy_edges, base_nbfnet.py:54-77: a filtered copy of the graph) and no plan is built per batch."""
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    cfg = {k: dict(v) for k, v in cfg.items()}
    cfg["entity_model_cfg"]["aggregate_func"] = aggr
    data = synthetic.make_kg(num_node=300, num_triple=2500, num_relation_base=5, num_test=16, seed=8).to(dev)
    torch.manual_seed(3)
    model = models.Ultra(**cfg)
    if aggr != "pna":                      # (pna's update layer has its own shape: random weights there)
        model.load_state_dict(state)
    model = model.to(dev).train()
    batch = torch.stack([data.edge_index[0, :4], data.edge_index[1, :4], data.edge_type[:4]], dim=-1)
    torch.manual_seed(0)
    neg = tasks.negative_sampling(data, batch, 16, strict=True)
    out = model(data, neg)
    n_plans = len(rspmm.cached_plans())
    loss = F.binary_cross_entropy_with_logits(out, torch.zeros_like(out))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    out2 = model(data, tasks.negative_sampling(data, batch.flip(0), 16, strict=True))       # another batch: same plans
    assert len(rspmm.cached_plans()) == n_plans and out2.shape == out.shape
    # the reference's route on the same batch
    h, t, r = neg.unbind(-1)
    filtered = model.entity_model.remove_easy_edges(data, h, t, r)
    assert filtered.edge_index.shape[1] < data.num_edges
    model.eval()
    with torch.no_grad():
        want = model(filtered, neg)
    assert torch.allclose(out.detach(), want, atol=(2e-4 if aggr == "pna" else 2e-5), rtol=1e-4)


def test_prefetched_negatives_equal_the_plain_loop_and_overlap_a_busy_stream(dev):
    """tasks.prefetch_negatives on the GPU: the draws of the plain loop (same generator order), handed over through the
    stream dependency -- checked while the training stream is kept busy between the batches, as a step would."""
    data = synthetic.make_kg(num_node=300, num_triple=2500, num_relation_base=5, num_test=16, seed=8).to(dev)
    triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)
    batches = [triples[8 * i:8 * i + 8] for i in range(6)]
    torch.manual_seed(11)
    plain = [tasks.negative_sampling(data, b, 32, strict=True) for b in batches]
    busy = torch.randn(2048, 2048, device=dev)
    torch.manual_seed(11)
    got = []
    for neg in tasks.prefetch_negatives(iter(batches), data, 32, strict=True):
        got.append(neg.clone())              # (read on the training stream: behind the side stream's event)
        for _ in range(4):
            busy = busy @ busy / 2048.0      # the "step": work queued on the training stream
    torch.cuda.synchronize()
    assert len(got) == len(plain) and all(torch.equal(a, b) for a, b in zip(got, plain))


def test_prefetched_negatives_behind_a_loader_that_builds_its_batches_with_kernels(dev):
    """script/run.py:32-34 keeps the triple list on the device: the DataLoader then stacks every batch with a KERNEL on the
    current stream.  The sampler's side stream must see that kernel's result, not the memory before it ran -- with the
    training stream kept busy for tens of milliseconds in front of every batch (the collate was queued behind the
    previous step and the sampler read the batch first)."""
    from torch.utils import data as torch_data
    data = synthetic.make_kg(num_node=300, num_triple=2500, num_relation_base=5, num_test=16, seed=8).to(dev)
    triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)[:64].contiguous()
    loader = torch_data.DataLoader(triples, 8)               # default collate: torch.stack of 8 GPU rows
    busy = torch.randn(4096, 4096, device=dev)
    for _ in range(6):
        busy = busy @ busy / 4096.0                           # the queue the first batch would wait behind
    torch.manual_seed(11)
    plain = [tasks.negative_sampling(data, triples[8 * i:8 * i + 8], 32, strict=True) for i in range(8)]
    torch.manual_seed(11)
    got = []
    for neg in tasks.prefetch_negatives(loader, data, 32, strict=True):
        got.append(neg.clone())
        for _ in range(6):
            busy = busy @ busy / 4096.0                       # the "step"
    torch.cuda.synchronize()
    assert len(got) == 8
    for i, (a, b) in enumerate(zip(got, plain)):
        assert torch.equal(a[:, 0], triples[8 * i:8 * i + 8]), "batch %d: the sampler read the positives before they were written" % i
        assert torch.equal(a, b)


def _index_add_rspmm(ei, et, rel, x, keep, n):
    msg = rel[:, et] * x[:, ei[1]]
    if keep is not None:
        msg = msg * keep.view(1, -1, 1)
    return torch.zeros(x.shape[0], n, x.shape[2], dtype=x.dtype, device=x.device).index_add(1, ei[0], msg)


@pytest.mark.parametrize("masked", [False, True])
def test_first_layer_rspmm_on_the_boundary_condition_matches_autograd(dev, masked):
    """rspmm.onehot_rspmm: layer 0's add_mul on its one-hot input (values[b] at row rows[b]) + that boundary -- output,
    relation gradient and values gradient against fp64 autograd of the dense formulation; one source is a hub, one has no
    out-edge, a keep mask drops a fifth of the edges."""
    gen = torch.Generator().manual_seed(21)
    n, e, bs, num_rel = 500, 6000, 4, 11
    ei = torch.randint(1, n, (2, e), generator=gen)          # (node 0 has no edge at all)
    ei[1, :1500] = 7                                         # hub source
    et = torch.randint(0, num_rel, (e,), generator=gen)
    rows = torch.tensor([7, 0, 123, 7])
    values = torch.randn(bs, 64, generator=gen)
    rel = torch.randn(bs, num_rel, 64, generator=gen)
    og = torch.randn(bs, n, 64, generator=gen)
    keep = (torch.rand(e, generator=gen) > 0.2).float() if masked else None

    v64, r64 = (t.double().to(dev).requires_grad_() for t in (values, rel))
    x0 = torch.zeros(bs, n, 64, dtype=torch.float64, device=dev).index_put((torch.arange(bs, device=dev), rows.to(dev)), v64)
    want = _index_add_rspmm(ei.to(dev), et.to(dev), r64, x0, keep.double().to(dev) if masked else None, n) + x0
    want.backward(og.double().to(dev))

    dei, det = ei.to(dev), et.to(dev)
    plan = rspmm.get_plan(dei, det, n, num_rel, exact_order=False)
    dv, dr = (t.to(dev).requires_grad_() for t in (values, rel))
    boundary = layers.PointBoundary(rows.to(dev), dv, n)
    out = rspmm.onehot_rspmm(plan, dei, det, dr, boundary.rows, dv, boundary.dense().detach(), keep.to(dev) if masked else None)
    torch.testing.assert_close(out.detach().double(), want.detach(), rtol=1e-5, atol=1e-5)
    out.backward(og.to(dev))
    torch.testing.assert_close(dr.grad.double(), r64.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dv.grad.double(), v64.grad, rtol=1e-4, atol=1e-3)


def test_closed_form_boundary_gradient_of_the_differentiable_rspmm(dev):
    """plan_rspmm(point=(rows, values)): same output and gradients as the boundary passed as a tensor built from
    `values`; the values' gradient is bs rows of the output gradient."""
    gen = torch.Generator().manual_seed(22)
    n, e, bs, num_rel = 400, 5000, 3, 9
    ei = torch.randint(0, n, (2, e), generator=gen).to(dev)
    et = torch.randint(0, num_rel, (e,), generator=gen).to(dev)
    rows = torch.tensor([5, 5, 399], device=dev)
    og = torch.randn(bs, n, 64, generator=gen).to(dev)
    plan = rspmm.get_plan(ei, et, n, num_rel, exact_order=False)
    grads = []
    for closed in (False, True):
        g2 = torch.Generator().manual_seed(23)
        values, rel, x = (t.to(dev).requires_grad_() for t in (torch.randn(bs, 64, generator=g2), torch.randn(bs, num_rel, 64, generator=g2),
                                                               torch.randn(bs, n, 64, generator=g2)))
        if closed:
            out = rspmm.plan_rspmm(plan, rel, x, point=(rows, values))
        else:
            dense_b = torch.zeros(bs, n, 64, device=dev).index_put((torch.arange(bs, device=dev), rows), values)
            out = rspmm.plan_rspmm(plan, rel, x, boundary=dense_b)
        out.backward(og)
        grads.append((out.detach(), values.grad, rel.grad, x.grad))
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("num_rel,n_edge", [(4, 40000), (37, 6000), (474, 6000), (600, 3000)])
def test_first_layer_backward_kernel_against_its_torch_restatement(dev, num_rel, n_edge):
    """csrc/onehot_bwd.hip (type-sorted runs per 16-lane group, boundary partials folded in order) against the padded-table
    torch formulation: hubs whose edges of one type span many groups, sources without edges, a relation table that fills
    the LDS (474) or exceeds it (600: declined, the torch route runs); the same bits run to run."""
    gen = torch.Generator().manual_seed(num_rel)
    n, bs = 600, 5
    ei = torch.randint(1, n, (2, n_edge), generator=gen)
    ei[1, : n_edge // 2] = 3                                 # hub: half of all edges leave node 3
    et = torch.randint(0, num_rel, (n_edge,), generator=gen)
    et[: n_edge // 4] = num_rel - 1                          # ... a quarter of them with one type (a run across many chunks)
    ei, et = ei.to(dev), et.to(dev)
    rows = torch.tensor([3, 0, 17, 3, 599], device=dev)
    values = torch.randn(bs, 64, generator=gen).to(dev)
    rel = torch.randn(bs, num_rel, 64, generator=gen).to(dev)
    og = torch.randn(bs, n, 64, generator=gen).to(dev)
    keep = (torch.rand(n_edge, generator=gen) > 0.3).float().to(dev)
    ptr, order, max_deg = rspmm.out_edge_csr(ei, et, n)
    for weight in (None, keep):
        want = rspmm._onehot_backward_torch(ptr, order, max_deg, ei, et, weight, rel.double(), rows, values.double(), og.double(),
                                            True, True)
        got = rspmm._onehot_backward_kernel(ptr, order, ei, et, weight, rel, rows, values, og, True, True)
        if num_rel * 256 + 64 * 2 * 68 * 4 > 160 * 1024:
            assert got is None
            continue
        torch.testing.assert_close(got[0].double(), want[0], rtol=1e-4, atol=2e-3)
        torch.testing.assert_close(got[1].double(), want[1], rtol=1e-4, atol=2e-3)
        again = rspmm._onehot_backward_kernel(ptr, order, ei, et, weight, rel, rows, values, og, True, True)
        assert torch.equal(got[0], again[0]) and torch.equal(got[1], again[1])
        only_rel = rspmm._onehot_backward_kernel(ptr, order, ei, et, weight, rel, rows, values, og, True, False)
        assert only_rel[1] is None and torch.equal(only_rel[0], got[0])


def test_tagged_edge_weights_are_permuted_once_and_never_stale(dev):
    """rspmm.tag_edge_weight: calls that carry the same tagged vector reuse the plan's permuted copy; an in-place write voids
    the tag, an untagged vector (even at the same address) is always permuted -- every result equals the untagged call's."""
    gen = torch.Generator().manual_seed(31)
    n, e, bs, num_rel = 300, 4000, 2, 6
    ei = torch.randint(0, n, (2, e), generator=gen).to(dev)
    et = torch.randint(0, num_rel, (e,), generator=gen).to(dev)
    rel = torch.randn(bs, num_rel, 64, generator=gen).to(dev)
    x = torch.randn(bs, n, 64, generator=gen).to(dev)
    plan = rspmm.Plan(ei, et, n, num_rel, exact_order=False, type_runs=False, dense=False)
    w = torch.rand(e, generator=gen).to(dev)
    want = plan.forward(rel, x, edge_weight=w.clone())
    rspmm.tag_edge_weight(w)
    assert rspmm._weight_epoch(w) > 0
    assert torch.equal(plan.forward(rel, x, edge_weight=w), want)
    assert torch.equal(plan.forward(rel, x, edge_weight=w), want)          # (the copy of the call before)
    w.mul_(2.0)                                                            # in place: the tag no longer applies
    assert rspmm._weight_epoch(w) == 0
    want2 = plan.forward(rel, x, edge_weight=w.clone())
    assert torch.equal(plan.forward(rel, x, edge_weight=w), want2) and not torch.equal(want2, want)
    rspmm.tag_edge_weight(w)
    assert torch.equal(plan.forward(rel, x, edge_weight=w), want2)
    w.copy_(torch.rand(e, generator=gen))                                  # same address, new contents, tag void
    assert torch.equal(plan.forward(rel, x, edge_weight=w), plan.forward(rel, x, edge_weight=w.clone()))
    # through autograd: forward and both backward walks with one tagged keep vector
    keep = rspmm.tag_edge_weight((torch.rand(e, generator=gen) > 0.3).float().to(dev))
    grads = []
    for tagged in (True, False):
        drel, dx = rel.clone().requires_grad_(), x.clone().requires_grad_()
        out = rspmm.plan_rspmm(plan, drel, dx, keep if tagged else keep.clone(), keep=True)
        out.sum().backward()
        grads.append((out.detach(), drel.grad, dx.grad))
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_training_layer_as_one_node_gives_the_two_nodes_gradients(dev):
    """dense.TrainLayerFunction (aggregate + update of a layer as one autograd node; the update's share of the input
    gradient is the base of the input-gradient walk) against the two nodes whose shares autograd adds: the same scores and
    the same parameter gradients, bit for bit (a + b = b + a), on a whole training step with a keep mask."""
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=300, num_triple=2500, num_relation_base=5, num_test=16, seed=8).to(dev)
    batch = torch.stack([data.edge_index[0, :4], data.edge_index[1, :4], data.edge_type[:4]], dim=-1)
    torch.manual_seed(0)
    neg = tasks.negative_sampling(data, batch, 16, strict=True)
    results = []
    was = layers.LAST_LAYER_ON_ROWS
    layers.LAST_LAYER_ON_ROWS = False         # (this test is about the whole-layer node: every layer takes it)
    for one_node in (True, False):
        layers.TRAINING_LAYER_NODE = one_node
        try:
            model = models.Ultra(**cfg)
            model.load_state_dict(state)
            model = model.to(dev).train()
            out = model(data, neg)
            F.binary_cross_entropy_with_logits(out, torch.zeros_like(out)).backward()
            results.append((out.detach(), {k: p.grad.clone() for k, p in model.named_parameters()}))
        finally:
            layers.TRAINING_LAYER_NODE = True
            if not one_node:
                layers.LAST_LAYER_ON_ROWS = was
    assert torch.equal(results[0][0], results[1][0])
    for name, g in results[0][1].items():
        assert torch.equal(g, results[1][1][name]), name


@pytest.mark.parametrize("bs,num_negative", [(8, 256), (5, 33), (2, 1)])
def test_strict_sampler_kernel_draws_the_reference_negatives(dev, bs, num_negative):
    """csrc/sampling.hip (the idx-th entity that is not a known answer, by bisection over the sorted answer keys) against
    the masks + nonzero() formulation of tasks.py:42-76 under the same generator state: the same (bs, 1 + k, 3) batch --
    repeated edges, a hub anchor with thousands of known answers, positives that are not edges of the graph."""
    data = synthetic.make_kg(num_node=400, num_triple=6000, num_relation_base=3, num_test=16, seed=12)
    ei, et = data.edge_index.clone(), data.edge_type.clone()
    ei[0, :2500] = 9                                       # node 9: ~ 2,500 edges as head, few relations -> hundreds of known tails
    data.edge_index, data.edge_type = ei, et
    data = data.to(dev)
    batch = torch.stack([data.edge_index[0, :bs], data.edge_index[1, :bs], data.edge_type[:bs]], dim=-1).clone()
    batch[-1, 1] = (batch[-1, 1] + 7) % 400                # a positive that (most likely) is no edge of the graph
    results = []
    for kernel in (True, False):
        tasks.STRICT_SAMPLER_KERNEL = kernel
        try:
            torch.manual_seed(77)
            results.append(tasks.negative_sampling(data, batch, num_negative, strict=True))
            results.append(tasks.negative_sampling(data, batch.flip(0), num_negative, strict=True))      # (generator carried on)
        finally:
            tasks.STRICT_SAMPLER_KERNEL = True
    assert torch.equal(results[0], results[2]) and torch.equal(results[1], results[3])
    t_mask, h_mask = tasks.strict_negative_mask(data, batch)
    out, half = results[0], bs // 2
    for i in range(half):
        assert t_mask[i, out[i, 1:, 1]].all()
    for i in range(half, bs):
        assert h_mask[i, out[i, 1:, 0]].all()


@pytest.mark.parametrize("message,masked", [("distmult", True), ("transe", False)])
def test_last_layer_on_listed_rows_matches_the_whole_layer(dev, message, masked):
    """dense.TrainRowsLayerFunction (the layer at the rows the readout reads: in-edges of the listed rows only, scatter
    backward) against the whole layer (dense.TrainLayerFunction) followed by a gather of the same rows: output and every
    gradient -- repeated rows, a hub row, an isolated row, the query's own row (point boundary), a keep mask."""
    gen = torch.Generator().manual_seed(41)
    n, e, bs, num_rel = 500, 7000, 3, 7
    ei = torch.randint(1, n, (2, e), generator=gen)
    ei[0, :1800] = 4                                      # hub row on the aggregation side
    et = torch.randint(0, num_rel, (e,), generator=gen)
    ei, et = ei.to(dev), et.to(dev)
    rows = torch.randint(1, n, (bs, 40), generator=gen)
    rows[:, 0] = 4                                        # the hub
    rows[:, 1] = rows[:, 2]                               # a repeated row
    rows[:, 3] = 0                                        # a row without in-edges
    point_rows = torch.tensor([4, 17, 0])
    rows[1, 5] = 17                                       # a listed row that carries the boundary value
    rows, point_rows = rows.to(dev), point_rows.to(dev)
    keep = (torch.rand(e, generator=gen) > 0.25).float().to(dev) if masked else None
    layer = layers.GeneralizedRelationalConv(64, 64, num_rel, 64, message_func=message, aggregate_func="sum", layer_norm=True,
                                             activation="relu", dependent=False).to(dev)
    plan = rspmm.get_plan(ei, et, n, num_rel, exact_order=False)
    flags = dense.CONV_LAYER_NORM | dense.CONV_RELU | dense.CONV_RESIDUAL
    og = torch.randn(bs, 40, 64, generator=gen).to(dev)
    results = []
    for on_rows in (True, False):
        g2 = torch.Generator().manual_seed(42)
        x, rel, values = (t.to(dev).requires_grad_() for t in (torch.randn(bs, n, 64, generator=g2), torch.randn(bs, num_rel, 64, generator=g2),
                                                               torch.randn(bs, 64, generator=g2)))
        layer.zero_grad()
        args = (layer.linear.weight, layer.linear.bias, layer.layer_norm.weight, layer.layer_norm.bias)
        mul = layer.message2mul[message]
        if on_rows:
            out = dense.TrainRowsLayerFunction.apply(plan, mul, 1e-5, flags, keep, rel, x, rows, None, point_rows, values, *args)
        else:
            whole = dense.TrainLayerFunction.apply(plan, mul, masked, 1e-5, flags, keep, rel, x, None, point_rows, values, *args)
            out = whole.gather(1, rows.unsqueeze(-1).expand(-1, -1, 64))
        out.backward(og)
        results.append([out.detach(), x.grad, rel.grad, values.grad] + [p.grad.clone() for p in args])
    for a, b in zip(*results):
        scale = max(b.abs().max().item(), 1.0)
        assert (a - b).abs().max().item() <= 2e-5 * scale, ((a - b).abs().max().item(), scale)


def test_training_step_with_the_last_layer_on_the_candidates_rows(dev):
    """A whole training step with layers.LAST_LAYER_ON_ROWS on / off: the same scores and parameter gradients up to fp32
    summation order (the listed rows are summed edge by edge, the whole layer by the plan's work items)."""
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=300, num_triple=2500, num_relation_base=5, num_test=16, seed=8).to(dev)
    batch = torch.stack([data.edge_index[0, :4], data.edge_index[1, :4], data.edge_type[:4]], dim=-1)
    torch.manual_seed(0)
    neg = tasks.negative_sampling(data, batch, 16, strict=True)
    results = []
    was = layers.LAST_LAYER_ON_ROWS
    for on_rows in (True, False):
        layers.LAST_LAYER_ON_ROWS = on_rows
        try:
            model = models.Ultra(**cfg)
            model.load_state_dict(state)
            model = model.to(dev).train()
            out = model(data, neg)
            assert model.entity_model._last_hidden_on_rows == on_rows
            F.binary_cross_entropy_with_logits(out, torch.zeros_like(out)).backward()
            results.append((out.detach(), {k: p.grad.clone() for k, p in model.named_parameters()}))
        finally:
            layers.LAST_LAYER_ON_ROWS = was
    assert torch.allclose(results[0][0], results[1][0], rtol=1e-4, atol=1e-5)
    for name, g in results[0][1].items():
        other = results[1][1][name]
        scale = max(other.abs().max().item(), 1e-6)
        assert (g - other).abs().max().item() <= 1e-4 * scale + 1e-7, name


def test_sampler_stream_runs_beside_the_training_stream(dev):
    """tasks.overlapping_stream: whatever streams the process made before (here: sixteen of both priorities, used), the stream the
    sampler gets runs a kernel to completion while the current stream is still busy."""
    used = []
    for i in range(16):
        s = torch.cuda.Stream(priority=-1 if i % 2 else 0)
        with torch.cuda.stream(s):
            torch.zeros(8, device=dev).add_(1)
        used.append(s)
    torch.cuda.synchronize()
    side = tasks.overlapping_stream(dev)
    assert side != torch.cuda.current_stream(dev)
    probe = torch.zeros(8, device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    busy, tiny = torch.cuda.Event(), torch.cuda.Event()
    torch.cuda._sleep(4_000_000)
    busy.record()
    with torch.cuda.stream(side):
        probe.add_(1)
        tiny.record(side)
    tiny.synchronize()
    assert not busy.query(), "the side stream's kernel waited for the current stream's"
    torch.cuda.synchronize()
    assert probe.tolist() == [1.0] * 8
