"""GeneralizedRelationalConv paths outside the BASELINE configs -- `pna` aggregation (layers.py:208-226), `rotate`
messages (the unfused PyG-semantics path, layers.py:135-181) and `dependent=True` relation features (layers.py:70-73) --
on the GPU against restatements of the reference's formulas on the CPU.

Parity status: the restatements below follow the reference file line by line, but `degree` / `scatter` come from
PyG / torch_scatter (not vendored in /root/reference), so these paths are PARITY UNPINNED against a live reference;
what is pinned is GPU == CPU restatement."""
import pytest
import torch
from torch.nn import functional as F

from oracle import rspmm_oracle
from ultra_amd import layers, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _graph(seed=3):
    kg = synthetic.make_kg(num_node=300, num_triple=2500, num_relation_base=6, num_test=8, seed=seed, relation_graph=False)
    return kg.edge_index, kg.edge_type, kg.num_nodes, kg.num_relations


def _pna_reference(layer, x, relation, boundary, edge_index, edge_type, mul):
    """layers.py:189-230 with aggregate_func == "pna", node-major like the reference."""
    bs, n = x.shape[:2]
    inp, rel, bnd = (t.transpose(0, 1).flatten(1) for t in (x, relation, boundary))
    ones = torch.ones(edge_index.shape[1])
    degree_out = (torch.bincount(edge_index[1], minlength=n).float() + 1).unsqueeze(-1)      # PyG degree(index) + 1, layers.py:193
    agg = lambda r, i, s: rspmm_oracle.generalized_rspmm(edge_index, edge_type, ones, r, i, sum=s, mul=mul)
    sum_, sq_sum = agg(rel, inp, "add"), agg(rel ** 2, inp ** 2, "add")
    max_, min_ = agg(rel, inp, "max"), agg(rel, inp, "min")
    mean = (sum_ + bnd) / degree_out
    sq_mean = (sq_sum + bnd ** 2) / degree_out
    max_, min_ = torch.max(max_, bnd), torch.min(min_, bnd)
    std = (sq_mean - mean ** 2).clamp(min=layer.eps).sqrt()
    features = torch.cat([mean.unsqueeze(-1), max_.unsqueeze(-1), min_.unsqueeze(-1), std.unsqueeze(-1)], dim=-1).flatten(-2)
    scale = degree_out.log()
    scale = scale / scale.mean()
    scales = torch.cat([torch.ones_like(scale), scale, 1 / scale.clamp(min=1e-2)], dim=-1)
    update = (features.unsqueeze(-1) * scales.unsqueeze(-2)).flatten(-2)
    update = update.view(n, bs, -1).transpose(0, 1)
    out = F.linear(torch.cat([x, update], dim=-1), layer.linear.weight, layer.linear.bias)     # layers.py:234-239
    if layer.layer_norm is not None:
        out = layer.layer_norm(out)
    return F.relu(out)


@pytest.mark.parametrize("message_func", ["distmult", "transe"])
def test_pna_layer_matches_the_restated_formulas(dev, message_func):
    ei, et, n, r = _graph()
    torch.manual_seed(1)
    layer = layers.GeneralizedRelationalConv(32, 32, r, 32, message_func, "pna", True, "relu")
    assert layer.linear.in_features == 13 * 32
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, n, 32, generator=g)
    bnd = torch.randn(3, n, 32, generator=g)
    query = torch.randn(3, 32, generator=g)
    with torch.no_grad():
        relation = layer.relation.weight.expand(3, -1, -1)
        want = _pna_reference(layer, x, relation, bnd, ei, et, layer.message2mul[message_func])
        glayer = layer.to(dev)
        got = glayer(x.to(dev), query.to(dev), bnd.to(dev), ei.to(dev), et.to(dev), (n, n)).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())


def test_dependent_relations_match_the_restated_formulas(dev):
    """dependent=True: relation features = relation_linear(query).view(bs, num_relation, dim) (layers.py:70-73)."""
    from oracle import ultra_oracle_model as om
    ei, et, n, r = _graph(seed=5)
    torch.manual_seed(4)
    layer = layers.GeneralizedRelationalConv(64, 64, r, 64, "distmult", "sum", True, "relu", dependent=True)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, n, 64, generator=g)
    bnd = torch.randn(2, n, 64, generator=g)
    query = torch.randn(2, 64, generator=g)
    with torch.no_grad():
        relation = layer.relation_linear(query).view(2, r, 64)
        sd = {"l." + k: v for k, v in layer.state_dict().items()}
        want = om.conv_layer(sd, "l.", x, relation, bnd, ei, et, n, "distmult", "sum", True)
        got = layer.to(dev)(x.to(dev), query.to(dev), bnd.to(dev), ei.to(dev), et.to(dev), (n, n)).cpu()
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())


def test_rotate_messages_take_the_unfused_path(dev):
    """rotate has no fused kernel (layers.py:91-94 routes it through message / aggregate / update with PyG semantics:
    gather edge_index[0], scatter to edge_index[1]): the GPU run of that torch code against its restatement on the CPU."""
    ei, et, n, r = _graph(seed=7)
    torch.manual_seed(8)
    layer = layers.GeneralizedRelationalConv(64, 64, r, 64, "rotate", "sum", True, "relu")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, n, 64, generator=g)
    bnd = torch.randn(2, n, 64, generator=g)
    query = torch.randn(2, 64, generator=g)
    with torch.no_grad():
        relation = layer.relation.weight.expand(2, -1, -1)
        # layers.py:135-181 restated: messages along edges + the boundary as self loops, summed into edge_index[1]
        x_j, r_j = x[:, ei[0]], relation[:, et]
        x_re, x_im = x_j.chunk(2, dim=-1)
        r_re, r_im = r_j.chunk(2, dim=-1)
        msg = torch.cat([x_re * r_re - x_im * r_im, x_re * r_im + x_im * r_re], dim=-1)
        msg = torch.cat([msg, bnd], dim=1)
        index = torch.cat([ei[1], torch.arange(n)])
        update = torch.zeros(2, n, 64).index_add_(1, index, msg)
        want = F.relu(layer.layer_norm(F.linear(torch.cat([x, update], dim=-1), layer.linear.weight, layer.linear.bias)))
        got = layer.to(dev)(x.to(dev), query.to(dev), bnd.to(dev), ei.to(dev), et.to(dev), (n, n)).cpu()
    assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
