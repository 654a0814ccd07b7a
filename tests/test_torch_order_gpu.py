"""The dense kernels of the layer follow the REFERENCE'S OPERATION ORDER (csrc/torch_math.hpp): nn.Linear as one
k-ascending fmaf chain per output with the bias added last, nn.LayerNorm as eight Welford accumulators merged in
order.  torch's CPU kernels compute exactly that (tests/test_torch_math.py), so the GPU results must equal
torch-on-CPU BIT FOR BIT -- layer outputs, hidden states after all six layers, and the readout MLP, whose last
128 -> 1 product is summed in the association of the host BLAS (ultra_amd/host_order.py probes it)."""
import pytest
import torch
from torch import nn
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


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


def _layer(seed, layer_norm=True):
    from ultra_amd import layers
    torch.manual_seed(seed)
    layer = layers.GeneralizedRelationalConv(64, 64, 4, 64, "distmult", "sum", layer_norm, "relu", dependent=False)
    with torch.no_grad():       # LayerNorm affine parameters away from their (1, 0) initialisation
        if layer_norm:
            layer.layer_norm.weight.normal_(1.0, 0.3)
            layer.layer_norm.bias.normal_(0.0, 0.3)
    return layer


def _reference_update(layer, x, agg, residual):
    out = F.linear(torch.cat([x, agg], dim=-1), layer.linear.weight, layer.linear.bias)      # layers.py:234-239
    if layer.layer_norm is not None:
        out = layer.layer_norm(out)
    out = F.relu(out)
    return out + x if residual else out


def test_host_torch_matches_the_restated_order():
    """The premise, checked on THIS host's CPU (the GPU box): torch == oracle/torch_math_oracle.c bit for bit."""
    from oracle import torch_math_oracle as tm
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 3000, 128, generator=g)
    w, b = torch.randn(64, 128, generator=g) * 0.2, torch.randn(64, generator=g)
    assert torch.equal(tm.linear(x, w, b), F.linear(x, w, b))
    y = torch.randn(4, 3000, 64, generator=g) * 3 + 1
    gm, bt = torch.randn(64, generator=g), torch.randn(64, generator=g)
    assert torch.equal(tm.layer_norm(y, gm, bt, 1e-5), F.layer_norm(y, (64,), gm, bt, 1e-5))


@pytest.mark.parametrize("rows", [1, 2, 31, 32, 33, 5000, 116328])
@pytest.mark.parametrize("layer_norm,residual", [(True, True), (True, False), (False, True)])
def test_conv_update_equals_torch_cpu_bitwise(dev, rows, layer_norm, residual):
    """Against the restated order (oracle/torch_math_oracle.c) at every size, and against torch's own CPU result where
    torch takes its regular GEMM path (a handful of rows go through MKL's remainder / GEMV kernels, which order their
    sums differently -- ULTRA's layers see batch * num_node rows)."""
    from oracle import torch_math_oracle as tm
    from ultra_amd import dense
    layer = _layer(rows, layer_norm)
    g = torch.Generator().manual_seed(rows + 1)
    x = torch.randn(1, rows, 64, generator=g) * 1.5
    agg = torch.randn(1, rows, 64, generator=g) * 4
    with torch.no_grad():
        want = tm.linear(torch.cat([x, agg], dim=-1), layer.linear.weight, layer.linear.bias)
        if layer_norm:
            want = tm.layer_norm(want, layer.layer_norm.weight, layer.layer_norm.bias, layer.layer_norm.eps)
        want = F.relu(want)
        want = want + x if residual else want
        want_torch = _reference_update(layer, x, agg, residual)
        got = dense.conv_update(layer.to(dev), x.to(dev), agg.to(dev), residual).cpu()
    assert torch.equal(got, want), "max |d| = %g, %d of %d elements differ" % (
        (got - want).abs().max().item(), int((got != want).sum()), got.numel())
    if rows >= 5000 or rows == 32:
        assert torch.equal(got, want_torch)


def test_relation_projection_equals_torch_cpu_bitwise(dev):
    from ultra_amd import dense
    torch.manual_seed(3)
    mlps = [nn.Sequential(nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 64)) for _ in range(6)]
    x = torch.randn(8, 474, 64)
    with torch.no_grad():
        want = torch.stack([m(x) for m in mlps])
        w0, b0 = torch.stack([m[0].weight for m in mlps]), torch.stack([m[0].bias for m in mlps])
        w2, b2 = torch.stack([m[2].weight for m in mlps]), torch.stack([m[2].bias for m in mlps])
        got = dense.relation_projection(x.to(dev), w0.to(dev).contiguous(), b0.to(dev).contiguous(),
                                        w2.to(dev).contiguous(), b2.to(dev).contiguous()).cpu()
    assert torch.equal(got, want), "max |d| = %g" % (got - want).abs().max().item()


def test_hidden_states_equal_the_reference_flow_bitwise(dev):
    """Ultra's whole propagation -- relation model, relation projections, six entity layers -- against the oracle's
    restatement of the reference's data flow (torch CPU ops + the reference-order rspmm): the final hidden states and
    the query vectors, bit for bit; the scores within the GEMV's rounding."""
    from oracle import ultra_oracle_model as om
    from ultra_amd import models, synthetic, tasks
    import os
    data = synthetic.make_kg(num_node=3000, num_triple=20000, num_relation_base=20, seed=5)
    cfg = synthetic.default_model_cfg()
    torch.manual_seed(0)
    model = models.Ultra(**cfg)
    golden = os.path.join(os.path.dirname(__file__), "golden", "ultra_3g_model.pt")
    model.load_state_dict(torch.load(golden))
    model.eval()
    batch = data.target_triples[:4]
    t_batch, h_batch = tasks.all_negative(data, batch)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    gmodel = model.to(dev)
    gdata = synthetic.to_device(data, dev)
    for cand in (t_batch, h_batch):
        captured = {}
        orig_gather = torch.Tensor.gather

        def spy_cat(tensors, dim=0, _cat=torch.cat):
            out = _cat(tensors, dim=dim)
            if len(tensors) == 2 and out.dim() == 3 and out.shape[-1] == 128 and out.shape[1] == data.num_nodes:
                captured["feature"] = out          # cat[hidden, node_query] of entity_nbfnet (models.py:166-170)
            return out
        om.torch.cat = spy_cat
        try:
            want = om.ultra_forward(sd, cfg, data, cand, rspmm_fn=om.reference_rspmm_fn())
        finally:
            om.torch.cat = torch.cat
        want_hidden = captured["feature"][..., :64]
        with torch.no_grad():
            got = gmodel(gdata, cand.to(dev)).cpu()
            h0, t0, r0 = gmodel.entity_model.negative_sample_to_tail(*cand.to(dev).unbind(-1), data.num_relations // 2)
            hiddens, _, query = gmodel.entity_model._bellmanford_hidden(gdata, h0[:, 0], r0[:, 0])
        assert torch.equal(hiddens[-1].cpu(), want_hidden), "hidden states differ: max |d| = %g" % (
            (hiddens[-1].cpu() - want_hidden).abs().max().item())
        assert torch.equal(query.cpu(), captured["feature"][:, 0, 64:])
        assert (got - want).abs().max().item() <= 5e-6


@pytest.mark.parametrize("bs,n,cand", [(8, 14541, None), (3, 500, 257), (1, 40, None)])
def test_readout_equals_torch_cpu_bitwise(dev, bs, n, cand):
    """score = mlp.2(relu(mlp.0(cat[hidden, query]))): mlp.0 as one k-ascending chain through both halves of the
    concatenated feature, mlp.2 in the host BLAS's association.  The kernel equals the restatement bit for bit on every
    row; torch itself equals it on all rows but the handful its BLAS sums with a remainder kernel."""
    import numpy as np
    from oracle import torch_math_oracle as tm
    from ultra_amd import dense, host_order, models, synthetic
    torch.manual_seed(7 + bs)
    net = models.EntityNBFNet(**{k: v for k, v in synthetic.default_model_cfg()["entity_model_cfg"].items() if k != "class"})
    g = torch.Generator().manual_seed(n)
    hidden = torch.randn(bs, n, 64, generator=g).relu()
    query = torch.randn(bs, 64, generator=g)
    t_index = torch.arange(n).unsqueeze(0).expand(bs, -1).contiguous() if cand is None \
        else torch.randint(0, n, (bs, cand), generator=g)
    with torch.no_grad():
        feature = torch.cat([hidden, query.unsqueeze(1).expand(-1, n, -1)], dim=-1)
        feature = feature.gather(1, t_index.unsqueeze(-1).expand(-1, -1, 128))
        want_torch = net.mlp(feature).squeeze(-1)
        hid = tm.linear(feature, net.mlp[0].weight, net.mlp[0].bias).relu().reshape(-1, 128)
        stages, source = host_order.readout_stages(128)
        last = host_order.emulate(stages, hid.numpy(), net.mlp[2].weight[0].numpy()).astype(np.float64)
        restated = torch.from_numpy((last + float(net.mlp[2].bias)).astype(np.float32)).view_as(want_torch)
        gnet = net.to(dev)
        got = dense.readout(gnet, hidden.to(dev), query.to(dev), t_index.to(dev)).cpu()
    assert torch.equal(got, restated)           # the kernel executes the program exactly, whatever the program is
    same = (got == want_torch).float().mean().item()
    assert (got - want_torch).abs().max().item() <= 4e-6
    # torch itself: all rows but the few its BLAS sums with a remainder kernel at the end of each thread's share (a host whose
    # tree is outside the family runs the ascending chain: closeness only)
    if source.startswith("host BLAS") and got.numel() >= 100000:
        assert same >= 0.995, same
