"""Fused MFMA epilogues (ultra_conv_update / ultra_readout) vs the plain PyTorch fp32 op chains they replace."""
import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _layer(dev, layer_norm=True, seed=0):
    from ultra_amd.layers import GeneralizedRelationalConv
    torch.manual_seed(seed)
    layer = GeneralizedRelationalConv(64, 64, 4, 64, "distmult", "sum", layer_norm, "relu").to(dev)
    if layer_norm:   # non-trivial affine parameters
        with torch.no_grad():
            layer.layer_norm.weight.uniform_(0.5, 1.5)
            layer.layer_norm.bias.uniform_(-0.5, 0.5)
    return layer


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 1000, 14541 * 8])
@pytest.mark.parametrize("layer_norm", [True, False])
@pytest.mark.parametrize("residual", [True, False])
def test_conv_update_matches_torch(dev, rows, layer_norm, residual):
    from ultra_amd import dense
    layer = _layer(dev, layer_norm)
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, 64, generator=g) * 2).to(dev)
    agg = (torch.randn(rows, 64, generator=g) * 5).to(dev)
    with torch.no_grad():
        assert dense.conv_update_supported(layer, x, agg)
        got = dense.conv_update(layer, x, agg, residual)
        want = layer.linear(torch.cat([x, agg], dim=-1))
        if layer_norm:
            want = layer.layer_norm(want)
        want = F.relu(want)
        if residual:
            want = want + x
        # and an fp64 restatement to tell which of the two fp32 results is closer to the truth
        l64 = torch.nn.Linear(128, 64).double().to(dev)
        l64.weight.copy_(layer.linear.weight.double())
        l64.bias.copy_(layer.linear.bias.double())
        t = l64(torch.cat([x, agg], dim=-1).double())
        if layer_norm:
            t = F.layer_norm(t, (64,), layer.layer_norm.weight.double(), layer.layer_norm.bias.double(), layer.layer_norm.eps)
        t = F.relu(t)
        if residual:
            t = t + x.double()
    err = (got - want).abs().max().item()
    assert err <= 2e-5, "max |fused - torch| = %g" % err
    e_got, e_want = (got.double() - t).abs().max().item(), (want.double() - t).abs().max().item()
    assert e_got <= 2 * e_want + 1e-6, "fused is further from the fp64 result than torch: %g vs %g" % (e_got, e_want)


def test_conv_update_asymmetric_weight_layout(dev):
    """Transpose / fragment-order detector: W = one-hot rows pick single input columns."""
    from ultra_amd import dense
    layer = _layer(dev, layer_norm=False)
    layer.activation = None
    with torch.no_grad():
        layer.linear.weight.zero_()
        layer.linear.bias.zero_()
        for f in range(64):
            layer.linear.weight[f, (37 * f + 5) % 128] = 1.0 + f
        x = torch.arange(33 * 64, dtype=torch.float32, device=dev).view(33, 64) / 7
        agg = -torch.arange(33 * 64, dtype=torch.float32, device=dev).view(33, 64) / 3
        got = dense.conv_update(layer, x, agg, residual=False)
        want = layer.linear(torch.cat([x, agg], dim=-1))
    assert torch.equal(got, want)


@pytest.mark.parametrize("bs,n,cand", [(1, 40, None), (3, 100, None), (8, 14541, None), (4, 300, 9), (2, 50, 257),
                                       (32, 70, None), (33, 70, None), (70, 20, 5)])
def test_readout_matches_torch(dev, bs, n, cand):
    from ultra_amd import dense, models, synthetic
    torch.manual_seed(bs)
    net = models.EntityNBFNet(**{k: v for k, v in synthetic.default_model_cfg()["entity_model_cfg"].items() if k != "class"}).to(dev)
    g = torch.Generator().manual_seed(n)
    hidden = torch.randn(bs, n, 64, generator=g).to(dev)
    query = torch.randn(bs, 64, generator=g).to(dev)
    if cand is None:
        t_index = torch.arange(n).unsqueeze(0).expand(bs, -1).contiguous().to(dev)
    else:
        t_index = torch.randint(0, n, (bs, cand), generator=g).to(dev)
    with torch.no_grad():
        assert dense.readout_supported(net, hidden)
        got = dense.readout(net, hidden, query, t_index)
        feature = torch.cat([hidden, query.unsqueeze(1).expand(-1, n, -1)], dim=-1)
        feature = feature.gather(1, t_index.unsqueeze(-1).expand(-1, -1, 128))
        want = net.mlp(feature).squeeze(-1)
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= 2e-5, "max |fused - torch| = %g" % err


def test_readout_flags_candidate_ids_outside_the_graph(dev):
    """An id outside [0, num_node) must not read out of bounds: its score is NaN, every other score is unchanged (the
    reference's gather raises an IndexError there, models.py:204-205)."""
    from ultra_amd import dense, models, synthetic
    torch.manual_seed(0)
    net = models.EntityNBFNet(**{k: v for k, v in synthetic.default_model_cfg()["entity_model_cfg"].items() if k != "class"}).to(dev)
    g = torch.Generator().manual_seed(1)
    hidden, query = torch.randn(3, 50, 64, generator=g).to(dev), torch.randn(3, 64, generator=g).to(dev)
    t_index = torch.randint(0, 50, (3, 40), generator=g).to(dev)
    bad = t_index.clone()
    bad[0, 3], bad[2, 39], bad[1, 0] = 50, -1, 1 << 40
    with torch.no_grad():
        good = dense.readout(net, hidden, query, t_index)
        got = dense.readout(net, hidden, query, bad)
    torch.cuda.synchronize()
    flagged = torch.zeros_like(got, dtype=torch.bool)
    flagged[0, 3] = flagged[2, 39] = flagged[1, 0] = True
    assert torch.isnan(got[flagged]).all() and torch.equal(got[~flagged], good[~flagged])


@pytest.mark.parametrize("bs,rows", [(1, 1), (3, 31), (8, 474), (2, 129)])
def test_relation_projection_matches_the_per_layer_modules(dev, bs, rows):
    """ultra_relation_projection == relation_projection (layers.py:80) of every entity layer, in one launch."""
    from ultra_amd import dense, models, synthetic
    torch.manual_seed(rows)
    net = models.EntityNBFNet(**{k: v for k, v in synthetic.default_model_cfg()["entity_model_cfg"].items() if k != "class"}).to(dev)
    rel = torch.randn(bs, rows, 64, generator=torch.Generator().manual_seed(bs)).to(dev)
    with torch.no_grad():
        net.query = rel
        got = net._project_relations_batched()
        want = [layer.relation_projection(rel) for layer in net.layers]
    assert len(got) == len(want) == 6
    for g, w in zip(got, want):
        assert g.shape == w.shape
        err = (g - w).abs().max().item()
        assert err <= 2e-5 * max(1.0, w.abs().max().item()), "max |fused - torch| = %g" % err


def test_relation_projection_weight_layout(dev):
    """Fragment-order detector: permutation-like weights make every output a single, identifiable input."""
    from ultra_amd import dense
    n_layer = 2
    w0 = torch.zeros(n_layer, 64, 64, device=dev)
    w2 = torch.zeros(n_layer, 64, 64, device=dev)
    for l in range(n_layer):
        for f in range(64):
            w0[l, f, (5 * f + 3 + l) % 64] = 1.0 + f          # hidden[f] = (1 + f) * x[(5 f + 3 + l) % 64]
            w2[l, f, (11 * f + 7 * l + 1) % 64] = 2.0 + l      # out[f] = (2 + l) * hidden[(11 f + 7 l + 1) % 64]
    b0 = torch.zeros(n_layer, 64, device=dev)
    b2 = torch.arange(n_layer * 64, dtype=torch.float32, device=dev).view(n_layer, 64)
    x = (torch.arange(40 * 64, dtype=torch.float32, device=dev).view(40, 64) % 97) / 8 + 0.125      # positive: relu is the identity
    got = dense.relation_projection(x, w0, b0, w2, b2)
    want = torch.stack([torch.relu(x @ w0[l].t() + b0[l]) @ w2[l].t() + b2[l] for l in range(n_layer)])
    assert torch.equal(got, want)


@pytest.mark.parametrize("bs,n,rels", [(1, 5, 3), (8, 14541, 474), (5, 33, 12)])
def test_query_boundary_matches_gather_and_scatter(dev, bs, n, rels):
    """models.py:131-141: query = rel_repr[arange, r]; boundary = zeros.scatter_add(head row, query)."""
    from ultra_amd import dense
    g = torch.Generator().manual_seed(n)
    table = torch.randn(bs, rels, 64, generator=g).to(dev)
    h = torch.randint(0, n, (bs,), generator=g).to(dev)
    r = torch.randint(0, rels, (bs,), generator=g).to(dev)
    boundary, query, qbias = dense.query_boundary(h, table, r, n)
    want_q = table[torch.arange(bs, device=dev), r]
    want_b = torch.zeros(bs, n, 64, device=dev)
    want_b.scatter_add_(1, h.view(bs, 1, 1).expand(-1, -1, 64), want_q.unsqueeze(1))
    assert qbias is None and torch.equal(query, want_q) and torch.equal(boundary, want_b)
    # with the readout MLP: the same launch also emits mlp.0's query half (models.py:166-170)
    mlp = torch.nn.Sequential(torch.nn.Linear(128, 128), torch.nn.ReLU(), torch.nn.Linear(128, 1)).to(dev)
    with torch.no_grad():
        boundary2, query2, qbias = dense.query_boundary(h, table, r, n, readout_mlp=mlp)
        want_qb = torch.nn.functional.linear(want_q, mlp[0].weight[:, 64:], mlp[0].bias)
    assert torch.equal(query2, want_q) and torch.equal(boundary2, want_b)
    assert qbias.shape == (bs, 128) and (qbias - want_qb).abs().max().item() <= 1e-5


@pytest.mark.parametrize("n_cand", [1, 2, 1023, 1024, 4095, 4096, 4097, 14541])
def test_batch_prologue_matches_torch(dev, n_cand):
    """base_nbfnet.py:79-86 + the asserts of models.py:196-197, for candidate counts around the kernel's unroll blocks."""
    from ultra_amd import dense
    g = torch.Generator().manual_seed(n_cand)
    bs, num_direct = 6, 11
    batch = torch.randint(0, 50, (bs, n_cand, 3), generator=g)
    batch[:, :, 2] = batch[:, :1, 2] % num_direct            # one relation per row
    batch[0::2, :, 0] = batch[0::2, :1, 0]                   # even rows: tail candidates (shared head)
    batch[1::2, :, 1] = batch[1::2, :1, 1]                   # odd rows: head candidates (shared tail)
    if n_cand > 1:
        batch[1::2, -1, 0] = 49 - batch[1::2, 0, 0]          # make sure the heads of the odd rows really differ (last slot)
        batch[4, n_cand // 2, 2] += 1                        # row 4 breaks the shared-relation rule -> invalid
    _, h0, r0, side, valid = dense.batch_prologue(batch.to(dev), num_direct)
    same = (batch == batch[:, :1]).all(dim=1)
    is_t = same[:, 0]
    assert torch.equal(side.cpu().bool(), is_t)
    assert torch.equal(h0.cpu(), torch.where(is_t, batch[:, 0, 0], batch[:, 0, 1]))
    assert torch.equal(r0.cpu(), torch.where(is_t, batch[:, 0, 2], batch[:, 0, 2] + num_direct))
    assert torch.equal(valid.cpu().bool(), (same[:, 0] | same[:, 1]) & same[:, 2])
