"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/torch_math_oracle.c: nn.Linear and nn.LayerNorm in
the operation order of torch's CPU kernels (see the C file).  The product package `ultra_amd` never imports this."""
import ctypes
import os

import torch

from . import build_oracle

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(build_oracle.MATH_OUT):
            build_oracle.build()
        _lib = ctypes.CDLL(build_oracle.MATH_OUT)
        vp, l = ctypes.c_void_p, ctypes.c_long
        _lib.oracle_linear_seq_f32.argtypes = [vp, vp, vp, vp, l, l, l]
        _lib.oracle_layer_norm_f32.argtypes = [vp, vp, vp, vp, l, ctypes.c_int, ctypes.c_float]
    return _lib


def linear(x, weight, bias=None):
    """F.linear(x, weight, bias) for fp32 CPU tensors: one k-ascending fmaf chain per output, bias added last."""
    x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
    w = weight.contiguous().float()
    out = torch.empty(x2.shape[0], w.shape[0])
    b = bias.contiguous().float() if bias is not None else None
    lib().oracle_linear_seq_f32(x2.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr(),
                                x2.shape[0], x2.shape[1], w.shape[0])
    return out.view(*x.shape[:-1], w.shape[0])


def layer_norm(x, weight=None, bias=None, eps=1e-5):
    """F.layer_norm(x, (N,), weight, bias, eps) for fp32 CPU tensors, N % 8 == 0."""
    n = x.shape[-1]
    assert n % 8 == 0
    x2 = x.reshape(-1, n).contiguous().float()
    out = torch.empty_like(x2)
    lib().oracle_layer_norm_f32(x2.data_ptr(), weight.contiguous().data_ptr() if weight is not None else None,
                                bias.contiguous().data_ptr() if bias is not None else None, out.data_ptr(), x2.shape[0], n,
                                float(eps))
    return out.view(x.shape)
