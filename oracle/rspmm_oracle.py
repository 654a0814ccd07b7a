"""ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

ctypes front-end of oracle/rspmm_oracle.c (the plain-C restatement of
/root/reference/ultra/rspmm/source/rspmm.cpp) plus a restatement of the Python dispatcher
/root/reference/ultra/rspmm/rspmm.py:168-179.  The product package `ultra_amd` never imports this.
"""
import ctypes
import os

import torch

from . import build_oracle

SUMS = {"add": 0, "min": 1, "max": 2}
MULS = {"mul": 0, "add": 1}

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build_oracle.OUT
        if not os.path.exists(path):
            path = build_oracle.build()
        _lib = ctypes.CDLL(path)
    return _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _suffix(dtype):
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64, got %s" % dtype)


def sort_edges(edge_index, edge_type, edge_weight):
    """rspmm.py:175-177: key = node_in * (node_out.max() + 1) + node_out; order = key.argsort().

    The reference's argsort is unstable; a stable sort is one of its legal outcomes and makes the
    oracle order reproducible.
    """
    node_in, node_out = edge_index
    if node_in.numel() == 0:
        return edge_index, edge_type, edge_weight, torch.zeros(0, dtype=torch.long)
    key = node_in * (node_out.max() + 1) + node_out
    order = key.argsort(stable=True)
    return edge_index[:, order], edge_type[order], edge_weight[order], order


def rspmm_forward(edge_index, edge_type, edge_weight, relation, input, sum="add", mul="mul"):
    """Sorted edges in, like RSPMM*Function.forward (rspmm.py:15-26) -> rspmm_forward_cpu."""
    edge_index = edge_index.contiguous().long().cpu()
    edge_type = edge_type.contiguous().long().cpu()
    edge_weight = edge_weight.contiguous().cpu()
    relation = relation.contiguous().cpu()
    input = input.contiguous().cpu()
    nnz = edge_index.shape[1]
    num_row, dim = input.shape
    if nnz:
        assert (edge_index[0].diff() >= 0).all(), "Expect sorted `edge_index`"
    output = torch.empty_like(input)
    fn = getattr(lib(), "oracle_rspmm_forward_" + _suffix(input.dtype))
    fn(ctypes.c_int(SUMS[sum]), ctypes.c_int(MULS[mul]), _ptr(edge_index), _ptr(edge_type),
       _ptr(edge_weight), _ptr(relation), _ptr(input), _ptr(output),
       ctypes.c_int64(nnz), ctypes.c_int64(num_row), ctypes.c_int64(dim))
    return output


def rspmm_backward(edge_index, edge_type, edge_weight, relation, input, output, output_grad,
                   sum="add", mul="mul"):
    """rspmm_backward_cpu (rspmm.cpp:164-219) -> (weight_grad, relation_grad, input_grad)."""
    edge_index = edge_index.contiguous().long().cpu()
    edge_type = edge_type.contiguous().long().cpu()
    edge_weight = edge_weight.contiguous().cpu()
    relation = relation.contiguous().cpu()
    input = input.contiguous().cpu()
    output = output.contiguous().cpu()
    output_grad = output_grad.contiguous().cpu()
    nnz = edge_index.shape[1]
    num_row, dim = input.shape
    weight_grad = torch.zeros_like(edge_weight)
    relation_grad = torch.zeros_like(relation)
    input_grad = torch.zeros_like(input)
    fn = getattr(lib(), "oracle_rspmm_backward_" + _suffix(input.dtype))
    fn(ctypes.c_int(SUMS[sum]), ctypes.c_int(MULS[mul]), _ptr(edge_index), _ptr(edge_type),
       _ptr(edge_weight), _ptr(relation), _ptr(input), _ptr(output), _ptr(output_grad),
       _ptr(weight_grad), _ptr(relation_grad), _ptr(input_grad),
       ctypes.c_int64(nnz), ctypes.c_int64(num_row), ctypes.c_int64(relation.shape[0]),
       ctypes.c_int64(dim))
    return weight_grad, relation_grad, input_grad


def generalized_rspmm(edge_index, edge_type, edge_weight, relation, input, sum="add", mul="mul"):
    """rspmm.py:168-179 (forward only): validates the pair, sorts, applies."""
    if sum not in SUMS or mul not in MULS:
        raise ValueError("No generalized rspmm implementation found for summation `%s` and "
                         "multiplication `%s`" % (sum, mul))
    ei, et, ew, _ = sort_edges(edge_index.cpu(), edge_type.cpu(), edge_weight.cpu())
    return rspmm_forward(ei, et, ew, relation, input, sum=sum, mul=mul)
