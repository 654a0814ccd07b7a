import torch


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.get_default_dtype(), device=index.device)
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype, device=index.device))


def index_to_mask(index, size=None):
    size = int(index.max()) + 1 if size is None else size
    mask = torch.zeros(size, dtype=torch.bool, device=index.device)
    mask[index] = True
    return mask
