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


@pytest.mark.parametrize("bs,n,cand", [(1, 40, None), (3, 100, None), (8, 14541, None), (4, 300, 9), (2, 50, 257)])
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
