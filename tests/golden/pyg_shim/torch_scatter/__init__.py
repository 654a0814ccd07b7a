"""Test-only stand-in for torch_scatter (scatter, scatter_add) built on torch index ops."""
import torch


def _expand(index, src, dim):
    dim = dim % src.dim()
    shape = [1] * src.dim()
    shape[dim] = -1
    return index.view(shape).expand_as(src)


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim=dim, out=out, dim_size=dim_size, reduce="sum")


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    dim = dim % src.dim()
    if index.dim() == 1 and src.dim() > 1:
        index = _expand(index, src, dim)
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    if reduce in ("sum", "add"):
        return torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, index, src)
    if reduce == "mean":
        total = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, index, src)
        count = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, index, torch.ones_like(src))
        return total / count.clamp(min=1)
    if reduce in ("max", "min"):
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
        return out.scatter_reduce(dim, index, src, reduce="amax" if reduce == "max" else "amin", include_self=False)
    raise ValueError(reduce)
