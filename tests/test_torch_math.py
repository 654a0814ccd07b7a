"""Pins oracle/torch_math_oracle.c -- the operation order the GPU kernels follow for nn.Linear / nn.LayerNorm -- against
torch itself on this host's CPU: bit-equal on random inputs of the shapes ULTRA uses (layers.py:233-240,
models.py:121-128).  torch is the reference's third-party dependency here (torch 2.10.0; the order is MKL sgemm's /
ATen RowwiseMoments'); if a different torch build orders differently these tests say so."""
import pytest
import torch
from torch.nn import functional as F

from oracle import torch_math_oracle as tm


@pytest.mark.parametrize("shape,out_dim", [((8, 474, 128), 64), ((3, 1000, 64), 64), ((2, 500, 128), 128), ((7, 64), 64)])
def test_linear_is_a_k_ascending_fmaf_chain(shape, out_dim):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g) * 2
    w = torch.randn(out_dim, shape[-1], generator=g) * 0.2
    b = torch.randn(out_dim, generator=g)
    assert torch.equal(tm.linear(x, w, b), F.linear(x, w, b))
    assert torch.equal(tm.linear(x, w), F.linear(x, w))


@pytest.mark.parametrize("rows", [1, 17, 5000])
@pytest.mark.parametrize("scale,shift", [(1.0, 0.0), (30.0, 5.0), (1e-3, -2.0)])
def test_layer_norm_is_eight_welford_accumulators_merged_in_order(rows, scale, shift):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(4, rows, 64, generator=g) * scale + shift
    gamma = torch.randn(64, generator=g)
    beta = torch.randn(64, generator=g)
    assert torch.equal(tm.layer_norm(x, gamma, beta, 1e-5), F.layer_norm(x, (64,), gamma, beta, 1e-5))
    assert torch.equal(tm.layer_norm(x), F.layer_norm(x, (64,)))
    const = torch.full((2, 64), 0.37)      # zero variance: rstd = 1 / sqrt(eps)
    assert torch.equal(tm.layer_norm(const, gamma, beta), F.layer_norm(const, (64,), gamma, beta))
