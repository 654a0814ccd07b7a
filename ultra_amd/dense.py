"""Host side of the fused dense epilogues (include/ultra_nbfnet.h): the MFMA kernels that replace the
torch op chains `cat -> Linear -> LayerNorm -> ReLU (-> + residual)` (layers.py:233-240, models.py:158-160)
and `cat -> gather -> Linear -> ReLU -> Linear` (models.py:166-170, 202-209) on the inference path."""
import ctypes

import torch
from torch.autograd.function import once_differentiable
from torch.nn import functional as F

from . import _lib as _lib_codes
from ._lib import check, lib

CONV_LAYER_NORM, CONV_RELU, CONV_RESIDUAL = 1, 2, 4
# the last layer's backward on the listed rows as gathers (A/B switch for tests: the scatter with float atomics is the other side)
ROWS_BACKWARD_GATHER = True


def _stream(t):
    """The current HIP stream of the operand's device (the C entry points make that device current for the launch)."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def conv_update_supported(layer, input, update):
    """The fused update kernels serve the ULTRA shape (64 -> 64, fp32); with gradients enabled through
    ConvUpdateFunction (forward kernel + ultra_conv_update_backward)."""
    return (input.is_cuda and input.dtype == torch.float32 and update.dtype == torch.float32
            and layer.input_dim == 64 and layer.output_dim == 64 and layer.linear.in_features == 128
            and update.shape[-1] == 64 and (layer.activation is None or layer.activation is F.relu)
            and input.numel() > 0
            and (layer.layer_norm is None or (layer.layer_norm.elementwise_affine and layer.layer_norm.bias is not None)))


def _ptr(t):
    return t.data_ptr() if t is not None else None


def pick_rows(table, rows):
    """table[b, rows[b], :] for every b -- table[arange(batch), rows] as ONE gather launch (no arange; under autograd its backward
    is a zero fill + scatter_add of distinct rows, where advanced indexing's index_put_ sorts its indices first)."""
    return table.gather(1, rows.view(-1, 1, 1).expand(-1, 1, table.shape[-1])).squeeze(1)


def _conv_update_forward(x, agg, weight, bias, ln_w, ln_b, eps, flags):
    out = torch.empty_like(x)
    check(lib.ultra_conv_update(x.data_ptr(), agg.data_ptr(), weight.data_ptr(), _ptr(bias), _ptr(ln_w), _ptr(ln_b),
                                out.data_ptr(), x.numel() // 64, 64, 64, eps, flags, _stream(x)))
    return out


class ConvUpdateFunction(torch.autograd.Function):
    """out = [x +] relu(LayerNorm(W . cat[x, agg] + b)) as ONE autograd node (the reference builds five: cat, addmm,
    native_layer_norm, relu, add -- layers.py:233-240).  Saves x and agg only; the backward recomputes the
    pre-activation on the matrix cores (csrc/conv_update_bwd.hip)."""

    @staticmethod
    def forward(ctx, x, agg, weight, bias, ln_w, ln_b, eps, flags):
        x, agg, weight = x.contiguous(), agg.contiguous(), weight.contiguous()
        ctx.eps, ctx.flags = eps, flags
        ctx.save_for_backward(x, agg, weight, bias, ln_w, ln_b)
        return _conv_update_forward(x, agg, weight, bias, ln_w, ln_b, eps, flags)

    @staticmethod
    @once_differentiable      # (the backward kernels are not themselves differentiable)
    def backward(ctx, grad_out):
        x, agg, weight, bias, ln_w, ln_b = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        rows = x.numel() // 64
        gx, gagg, gw = torch.empty_like(x), torch.empty_like(agg), torch.empty_like(weight)
        gb = torch.empty_like(bias) if bias is not None else None
        gln_w = torch.empty_like(ln_w) if ln_w is not None else None
        gln_b = torch.empty_like(ln_b) if ln_b is not None else None
        nbytes = lib.ultra_conv_update_backward_workspace(rows)
        work = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        check(lib.ultra_conv_update_backward(x.data_ptr(), agg.data_ptr(), grad_out.data_ptr(), weight.data_ptr(), _ptr(bias),
                                             _ptr(ln_w), _ptr(ln_b), gx.data_ptr(), gagg.data_ptr(), gw.data_ptr(), _ptr(gb),
                                             _ptr(gln_w), _ptr(gln_b), work.data_ptr(), nbytes, rows, 64, 64, ctx.eps,
                                             ctx.flags, _stream(x)))
        return gx, gagg, gw, gb, gln_w, gln_b, None, None


class TrainLayerFunction(torch.autograd.Function):
    """A whole NBFNet layer of a training step as ONE autograd node (sum aggregate):

        agg = rspmm(relation, x, keep mask) + boundary;      out = [x +] relu(LayerNorm(W . cat[x, agg] + b))

    x feeds the rspmm AND the update.  As two nodes (rspmm._PlanRSPMM, ConvUpdateFunction) each returns its share of x's
    gradient and autograd adds them -- a pass over three (batch, N, 64) tensors per layer (13 us at FB15k237's size, 100 at
    YAGO3-10's); here the update's share is the base the input-gradient walk starts from (ultra_rspmm_backward_add).  The
    boundary is the closed form (rows, values) -- its gradient bs rows of the aggregate's -- or a tensor."""

    @staticmethod
    def forward(ctx, plan, mul, keep, eps, flags, edge_weight, relation, x, boundary, point_rows, point_values, weight, bias,
                ln_w, ln_b):
        from . import rspmm
        x, weight = x.contiguous(), weight.contiguous()
        point = (point_rows, point_values) if point_rows is not None else None
        agg = plan.forward(relation, x, edge_weight=edge_weight, boundary=boundary, sum="add", mul=mul, keep=keep, point=point)
        ctx.plan, ctx.mul, ctx.eps, ctx.flags, ctx.point_rows = plan, mul, eps, flags, point_rows
        ctx.weight_epoch = rspmm._weight_epoch(edge_weight)
        ctx.save_for_backward(edge_weight, relation, x, agg, weight, bias, ln_w, ln_b)
        return _conv_update_forward(x, agg, weight, bias, ln_w, ln_b, eps, flags)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        edge_weight, relation, x, agg, weight, bias, ln_w, ln_b = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        rows = x.numel() // 64
        gx, gagg, gw = torch.empty_like(x), torch.empty_like(agg), torch.empty_like(weight)
        gb = torch.empty_like(bias) if bias is not None else None
        gln_w = torch.empty_like(ln_w) if ln_w is not None else None
        gln_b = torch.empty_like(ln_b) if ln_b is not None else None
        nbytes = lib.ultra_conv_update_backward_workspace(rows)
        work = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        check(lib.ultra_conv_update_backward(x.data_ptr(), agg.data_ptr(), grad_out.data_ptr(), weight.data_ptr(), _ptr(bias),
                                             _ptr(ln_w), _ptr(ln_b), gx.data_ptr(), gagg.data_ptr(), gw.data_ptr(), _ptr(gb),
                                             _ptr(gln_w), _ptr(gln_b), work.data_ptr(), nbytes, rows, 64, 64, ctx.eps,
                                             ctx.flags, _stream(x)))
        need = ctx.needs_input_grad
        relation_grad, x_grad = None, gx
        if need[6] or need[7]:
            epoch = ctx.weight_epoch if (edge_weight is not None and edge_weight.is_contiguous()) else 0
            _, relation_grad, x_grad = ctx.plan.backward(relation, x, agg, gagg, edge_weight=edge_weight, sum="add", mul=ctx.mul,
                                                         weight_epoch=epoch, input_grad_base=gx)
        values_grad = None
        if ctx.point_rows is not None and need[10]:
            r = ctx.point_rows
            values_grad = pick_rows(gagg, r)
        return (None, None, None, None, None, None, relation_grad if need[6] else None, x_grad if need[7] else None,
                gagg if need[8] else None, None, values_grad, gw, gb, gln_w, gln_b)


class TrainRowsLayerFunction(torch.autograd.Function):
    """The LAST layer of a training step, evaluated at the rows the readout reads (models.py:202-207 gathers the candidates'
    rows of the last hidden state: 1 + num_negative of N per sample):

        out[b, j] = update(x[b, rows[b, j]], sum over the in-edges of rows[b, j] of w rel (x) x[col] + boundary[b, rows[b, j]])

    -- the in-edges of ~ 2,000 listed rows instead of every edge of the graph, forward (ultra_rspmm_rows_forward) and
    backward (ultra_rspmm_rows_backward: a scatter with float atomics into the zeroed input / relation gradients, as the
    reference's GPU backward does), and the update and its backward on bs x (1 + num_negative) rows instead of bs x N.
    Returns (batch, n_list, 64).  Boundary: the closed form (point_rows, point_values) or a tensor without gradient."""

    @staticmethod
    def forward(ctx, plan, mul, eps, flags, edge_weight, relation, x, rows, boundary, point_rows, point_values, weight, bias,
                ln_w, ln_b):
        from . import rspmm
        x, weight = x.contiguous(), weight.contiguous()
        rows = rows.to(torch.int64).contiguous()
        bs, n_list = rows.shape
        relation_c, mrel = rspmm.as_mat(relation)
        _, mx = rspmm.as_mat(x)
        mb = None
        if boundary is not None:
            boundary, mbv = rspmm.as_mat(boundary)
            mb = ctypes.byref(mbv)
        if point_values is not None:
            point_values = point_values.contiguous()
        agg = torch.empty(bs, n_list, 64, dtype=torch.float32, device=x.device)
        w = edge_weight.contiguous() if edge_weight is not None else None
        check(lib.ultra_rspmm_rows_forward(plan._h, rspmm._lib.MUL_CODES[mul], _ptr(w), ctypes.byref(mrel), ctypes.byref(mx),
                                           rows.data_ptr(), n_list, mb, _ptr(point_rows), _ptr(point_values), agg.data_ptr(),
                                           _stream(x)))
        x_rows = x.gather(1, rows.unsqueeze(-1).expand(-1, -1, 64))
        ctx.plan, ctx.mul, ctx.eps, ctx.flags = plan, mul, eps, flags
        ctx.save_for_backward(w, relation_c, x, rows, x_rows, agg, point_rows, weight, bias, ln_w, ln_b)
        return _conv_update_forward(x_rows, agg, weight, bias, ln_w, ln_b, eps, flags)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        from . import rspmm
        w, relation, x, rows, x_rows, agg, point_rows, weight, bias, ln_w, ln_b = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        n = x_rows.numel() // 64
        gx_rows, gagg, gw = torch.empty_like(x_rows), torch.empty_like(agg), torch.empty_like(weight)
        gb = torch.empty_like(bias) if bias is not None else None
        gln_w = torch.empty_like(ln_w) if ln_w is not None else None
        gln_b = torch.empty_like(ln_b) if ln_b is not None else None
        nbytes = lib.ultra_conv_update_backward_workspace(n)
        work = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        check(lib.ultra_conv_update_backward(x_rows.data_ptr(), agg.data_ptr(), grad_out.data_ptr(), weight.data_ptr(), _ptr(bias),
                                             _ptr(ln_w), _ptr(ln_b), gx_rows.data_ptr(), gagg.data_ptr(), gw.data_ptr(), _ptr(gb),
                                             _ptr(gln_w), _ptr(gln_b), work.data_ptr(), nbytes, n, 64, 64, ctx.eps,
                                             ctx.flags, _stream(x)))
        need = ctx.needs_input_grad
        relation_grad = x_grad = values_grad = None
        want_values = point_rows is not None and need[10]
        if need[5] or need[6]:
            _, mrel = rspmm.as_mat(relation)
            _, mx = rspmm.as_mat(x)
            rc = _lib_codes.ULTRA_ERR_UNSUPPORTED
            if ROWS_BACKWARD_GATHER:
                # both gradients as gathers, written in full: no zero fill, no scatter of the update's share, no atomics
                x_grad = torch.empty_like(x)
                relation_grad = torch.empty(relation.shape, dtype=torch.float32, device=x.device)
                _, mrg = rspmm.as_mat(relation_grad)
                _, mxg = rspmm.as_mat(x_grad)
                if want_values:
                    values_grad = torch.empty(rows.shape[0], 64, dtype=torch.float32, device=x.device)
                rc = lib.ultra_rspmm_rows_backward_gather(ctx.plan._h, rspmm._lib.MUL_CODES[ctx.mul], _ptr(w), ctypes.byref(mrel),
                                                          ctypes.byref(mx), rows.data_ptr(), rows.shape[1], gagg.data_ptr(),
                                                          gx_rows.data_ptr(), _ptr(point_rows) if want_values else None,
                                                          _ptr(values_grad), ctypes.byref(mrg), ctypes.byref(mxg), _stream(x))
                if rc != _lib_codes.ULTRA_ERR_UNSUPPORTED:
                    check(rc)
            if rc == _lib_codes.ULTRA_ERR_UNSUPPORTED:
                # the scatter with float atomics (round 5): the update's share of the input gradient lands on the listed rows
                # (repeats add up); the rspmm's is scattered on top
                values_grad = None
                x_grad = torch.zeros_like(x).scatter_add_(1, rows.unsqueeze(-1).expand(-1, -1, 64), gx_rows)
                relation_grad = torch.zeros(relation.shape, dtype=torch.float32, device=x.device)
                _, mrg = rspmm.as_mat(relation_grad)
                _, mxg = rspmm.as_mat(x_grad)
                check(lib.ultra_rspmm_rows_backward(ctx.plan._h, rspmm._lib.MUL_CODES[ctx.mul], _ptr(w), ctypes.byref(mrel),
                                                    ctypes.byref(mx), rows.data_ptr(), rows.shape[1], gagg.data_ptr(),
                                                    ctypes.byref(mrg), ctypes.byref(mxg), _stream(x)))
        if want_values and values_grad is None:
            hit = (rows == point_rows.unsqueeze(1)).to(gagg.dtype).unsqueeze(-1)
            values_grad = (gagg * hit).sum(dim=1)
        return (None, None, None, None, None, relation_grad if need[5] else None, x_grad if need[6] else None, None, None, None,
                values_grad, gw, gb, gln_w, gln_b)


def conv_update(layer, input, update, residual):
    """out = [input +] relu(layer_norm(linear(cat[input, update]))) for (..., 64) fp32 GPU tensors."""
    flags = (CONV_LAYER_NORM if layer.layer_norm is not None else 0) | (CONV_RELU if layer.activation is not None else 0) \
        | (CONV_RESIDUAL if residual else 0)
    ln = layer.layer_norm
    eps = float(ln.eps) if ln is not None else 1e-5
    args = (input, update, layer.linear.weight, layer.linear.bias, ln.weight if ln is not None else None,
            ln.bias if ln is not None else None)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in args):
        return ConvUpdateFunction.apply(*args, eps, flags)
    return _conv_update_forward(input.contiguous(), update.contiguous(), layer.linear.weight, layer.linear.bias, args[4],
                                args[5], eps, flags)


EDGE_KEEP_MAX_EASY = 8192


def edge_keep_mask(edge_index, edge_type, easy_edge, num_node, num_relation, dtype=torch.float32):
    """keep[e] = 0 where graph edge e equals one of the columns of easy_edge ((3, M): heads, tails, types; (2, M) with
    edge_type=None: heads, tails), else 1 -- the edge dropout of base_nbfnet.py:54-77 as a 0/1 vector: one small sort of
    the M easy keys and one kernel over the edges, instead of sorting the graph's edge keys every batch."""
    assert edge_index.dtype == torch.int64 and edge_index.is_cuda
    assert int(num_node) ** 2 * max(int(num_relation), 1) < 2 ** 63, "edge key overflows int64"      # (tasks.py:19, same bound)
    edge_index = edge_index.contiguous()
    head, tail = edge_index[0], edge_index[1]
    easy_edge = easy_edge.to(torch.int64)
    key = easy_edge[0] * int(num_node) + easy_edge[1]
    if edge_type is not None:
        edge_type = edge_type.contiguous()
        key = key * int(num_relation) + easy_edge[2]
    key = key.sort()[0].contiguous()
    keep = torch.empty(edge_index.shape[1], dtype=torch.float32, device=edge_index.device)
    check(lib.ultra_edge_keep_mask(head.data_ptr(), tail.data_ptr(), _ptr(edge_type), edge_index.shape[1], key.data_ptr(),
                                   key.numel(), int(num_node), int(num_relation), keep.data_ptr(), _stream(keep)))
    return keep if dtype == torch.float32 else keep.to(dtype)


def easy_edge_keep(edge_index, edge_type, h_index, t_index, r_index, num_node, num_relation, dtype=torch.float32):
    """edge_keep_mask straight from the batch's (h, t, r) -- the list of base_nbfnet.py:57-59 (every triple and its inverse) is
    never built: one launch (ultra_easy_edge_keep).  h / t / r: equally shaped int64 GPU tensors that are either contiguous
    or the columns of one contiguous (..., 3) tensor; None where that does not hold (the caller builds the list)."""
    n = h_index.numel()
    if not (edge_index.is_cuda and edge_index.dtype == torch.int64 and h_index.dtype == torch.int64 and h_index.is_cuda
            and 0 < 2 * n <= EDGE_KEEP_MAX_EASY and h_index.shape == t_index.shape == r_index.shape
            and int(num_node) ** 2 * max(int(num_relation), 1) < 2 ** 62):
        return None
    if h_index.is_contiguous() and t_index.is_contiguous() and r_index.is_contiguous():
        stride = 1
    else:
        # columns of a contiguous (..., 3) tensor: element i of each sits 3 i elements behind its first
        want = tuple(3 * s for s in torch.empty(h_index.shape, device="meta").stride())
        if not (h_index.stride() == t_index.stride() == r_index.stride() == want):
            return None
        stride = 3
    edge_index = edge_index.contiguous()
    if edge_type is not None:
        edge_type = edge_type.contiguous()
    keep = torch.empty(edge_index.shape[1], dtype=torch.float32, device=edge_index.device)
    check(lib.ultra_easy_edge_keep(edge_index[0].data_ptr(), edge_index[1].data_ptr(), _ptr(edge_type), edge_index.shape[1],
                                   h_index.data_ptr(), t_index.data_ptr(), r_index.data_ptr() if edge_type is not None else None,
                                   n, stride, int(num_node), int(num_relation), int(num_relation) // 2, keep.data_ptr(),
                                   _stream(keep)))
    return keep if dtype == torch.float32 else keep.to(dtype)


def readout_supported(model, hidden):
    mlp = model.mlp
    return (hidden.is_cuda and hidden.dtype == torch.float32 and hidden.shape[-1] == 64 and not model.concat_hidden
            and len(mlp) == 3 and isinstance(mlp[0], torch.nn.Linear) and isinstance(mlp[1], torch.nn.ReLU)
            and isinstance(mlp[2], torch.nn.Linear) and mlp[0].in_features == 128 and mlp[0].out_features == 128
            and mlp[2].out_features == 1 and mlp[0].bias is not None and mlp[2].bias is not None
            and mlp[0].weight.is_contiguous()
            and not (torch.is_grad_enabled() and (hidden.requires_grad or mlp[0].weight.requires_grad)))


READOUT_TRAIN_NODE = True


def readout_train_supported(model, hidden_rows, query):
    """The training step's readout as one autograd node (ReadoutTrainFunction): the ULTRA shape -- 64-d hidden rows of the
    candidates, mlp = Linear(128, 128), ReLU, Linear(128, 1) -- in fp32 on the GPU."""
    mlp = model.mlp
    return (READOUT_TRAIN_NODE and hidden_rows.is_cuda and hidden_rows.dtype == torch.float32 and hidden_rows.dim() == 3
            and hidden_rows.shape[-1] == 64 and query.shape == (hidden_rows.shape[0], 64) and query.dtype == torch.float32
            and hidden_rows.shape[0] <= 4096 and not model.concat_hidden
            and len(mlp) == 3 and isinstance(mlp[0], torch.nn.Linear) and isinstance(mlp[1], torch.nn.ReLU)
            and isinstance(mlp[2], torch.nn.Linear) and tuple(mlp[0].weight.shape) == (128, 128)
            and tuple(mlp[2].weight.shape) == (1, 128) and mlp[0].bias is not None and mlp[2].bias is not None
            and mlp[0].weight.dtype == torch.float32)


class ReadoutTrainFunction(torch.autograd.Function):
    """score = mlp(cat[hidden_rows, query]) (models.py:202-207 on the candidates' rows) with the ULTRA readout MLP: one launch
    forward (ultra_readout_train_forward), two backward (ultra_readout_train_backward) instead of torch's cat + two products +
    relu + copies and their dozen backward launches.  Sums in a fixed order: reproducible run to run."""

    @staticmethod
    def forward(ctx, hidden_rows, query, w1, b1, w2, b2):
        hidden_rows, query = hidden_rows.contiguous(), query.contiguous()
        w1, b1, w2, b2 = w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
        bs, n = hidden_rows.shape[:2]
        h = torch.empty(bs * n, 128, dtype=torch.float32, device=hidden_rows.device)
        score = torch.empty(bs, n, dtype=torch.float32, device=hidden_rows.device)
        check(lib.ultra_readout_train_forward(hidden_rows.data_ptr(), query.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                              b2.data_ptr(), h.data_ptr(), score.data_ptr(), bs, n, _stream(score)))
        ctx.save_for_backward(hidden_rows, query, w1, w2, h)
        return score

    @staticmethod
    def backward(ctx, grad_score):
        hidden_rows, query, w1, w2, h = ctx.saved_tensors
        grad_score = grad_score.contiguous()
        bs, n = hidden_rows.shape[:2]
        dev = hidden_rows.device
        g_hid, g_query = torch.empty_like(hidden_rows), torch.empty_like(query)
        g_w1, g_w2 = torch.empty_like(w1), torch.empty_like(w2)
        g_b1 = torch.empty(128, dtype=torch.float32, device=dev)
        g_b2 = torch.empty(1, dtype=torch.float32, device=dev)
        nbytes = lib.ultra_readout_train_backward_workspace(bs, n)
        work = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        check(lib.ultra_readout_train_backward(grad_score.data_ptr(), h.data_ptr(), hidden_rows.data_ptr(), query.data_ptr(),
                                               w1.data_ptr(), w2.data_ptr(), g_hid.data_ptr(), g_query.data_ptr(), g_w1.data_ptr(),
                                               g_b1.data_ptr(), g_w2.data_ptr(), g_b2.data_ptr(), work.data_ptr(), nbytes, bs, n,
                                               _stream(g_hid)))
        return g_hid, g_query, g_w1, g_b1, g_w2, g_b2


def readout_train(model, hidden_rows, query):
    mlp = model.mlp
    return ReadoutTrainFunction.apply(hidden_rows, query, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)


_ORDER_CACHE = {}
_ORDER_RETIRED = []      # programs replaced by host_order.adopt(): kept alive for captured graphs that still point at them


def readout_order(device):
    """The summation program of the readout's last product on `device` (host_order.readout_program: the association the
    host BLAS uses for nn.Linear(128, 1), probed once per process), as an int32 tensor."""
    key = str(device)
    if key not in _ORDER_CACHE:
        from . import host_order
        prog, _source = host_order.readout_program(128)
        _ORDER_CACHE[key] = torch.tensor(prog, dtype=torch.int32, device=device)
    return _ORDER_CACHE[key]


def readout(model, hidden, query, t_index, order=None):
    """score[b, i] = mlp(cat[hidden[b, t_index[b, i]], query[b]]) without materialising the concatenation, in the
    reference's operation order (csrc/dense_kernels.hip: readout_kernel)."""
    mlp = model.mlp
    hidden = hidden.contiguous()
    query = query.contiguous()
    batch, num_node = hidden.shape[:2]
    t_index = t_index.to(torch.int64).contiguous()      # the kernel reads int64 ids (an int32 batch would be misread)
    n_cand = t_index.shape[1]
    order = readout_order(hidden.device) if order is None else order
    score = torch.empty(batch, n_cand, dtype=hidden.dtype, device=hidden.device)
    check(lib.ultra_readout(hidden.data_ptr(), t_index.data_ptr(), mlp[0].weight.data_ptr(), query.data_ptr(),
                            mlp[0].bias.data_ptr(), mlp[2].weight.data_ptr(), mlp[2].bias.data_ptr(), order.data_ptr(),
                            order.numel(), score.data_ptr(), batch, num_node, n_cand, 64, 128, _stream(hidden)))
    return score


class Prologue(tuple):
    """(batch, h0, r0, side, valid) of batch_prologue, plus .rel_first: every row's relation as given (the relation model's
    query, models.py:20) -- Ultra.forward runs the prologue once, ahead of the relation model, and hands it on."""
    rel_first = None


def batch_prologue(batch, num_direct_rel, candidates=False):
    """(batch, h0, r0, side, valid) of a (bs, n_cand, 3) GPU batch in one kernel (models.py:190-197, base_nbfnet.py:79-86);
    candidates=True: also `.cand`, the (bs, n_cand) candidate nodes of the converted rows (new_t_index, base_nbfnet.py:84)."""
    batch = batch.contiguous()
    bs, n_cand = batch.shape[:2]
    cand = torch.empty(bs, n_cand, dtype=torch.long, device=batch.device) if candidates else None
    h0 = torch.empty(bs, dtype=torch.long, device=batch.device)
    r0 = torch.empty_like(h0)
    rel_first = torch.empty_like(h0)
    side = torch.empty(bs, dtype=torch.int32, device=batch.device)
    valid = torch.empty(bs, dtype=torch.int32, device=batch.device)
    check(lib.ultra_batch_prologue_rows(batch.data_ptr(), bs, n_cand, int(num_direct_rel), h0.data_ptr(), r0.data_ptr(),
                                        side.data_ptr(), valid.data_ptr(), rel_first.data_ptr(), _ptr(cand), _stream(batch)))
    out = Prologue((batch, h0, r0, side, valid))
    out.rel_first = rel_first
    out.cand = cand
    return out


def readout_batch(model, hidden, query, batch, side, order=None):
    """readout() with the candidate node read straight from the raw (bs, n_cand, 3) batch."""
    mlp = model.mlp
    hidden = hidden.contiguous()
    query = query.contiguous()
    bs, num_node = hidden.shape[:2]
    n_cand = batch.shape[1]
    order = readout_order(hidden.device) if order is None else order
    score = torch.empty(bs, n_cand, dtype=hidden.dtype, device=hidden.device)
    check(lib.ultra_readout_batch(hidden.data_ptr(), batch.data_ptr(), side.data_ptr(), mlp[0].weight.data_ptr(),
                                  query.data_ptr(), mlp[0].bias.data_ptr(), mlp[2].weight.data_ptr(), mlp[2].bias.data_ptr(),
                                  order.data_ptr(), order.numel(), score.data_ptr(), bs, num_node, n_cand, 64, 128,
                                  _stream(hidden)))
    return score


def boundary_supported(index, values):
    return (index.is_cuda and not (torch.is_grad_enabled() and values is not None and values.requires_grad)
            and (values is None or (values.dtype == torch.float32 and values.shape[-1] % 4 == 0)))


def onehot_boundary(index, values, num_node, dim):
    """(batch, num_node, dim) fp32 boundary: values[b] (or ones) at row index[b], zeros elsewhere -- one kernel."""
    batch = index.shape[0]
    out = torch.empty(batch, num_node, dim, dtype=torch.float32, device=index.device)
    index = index.to(torch.int64).contiguous()
    values = values.contiguous() if values is not None else None     # (kept referenced until the launch is enqueued)
    check(lib.ultra_onehot_rows(out.data_ptr(), index.data_ptr(), values.data_ptr() if values is not None else None, batch,
                                num_node, dim, _stream(out)))
    return out


def query_boundary(h_index, relation_representations, r_index, num_node, readout_mlp=None, materialize=True):
    """(boundary, query, qbias) of EntityNBFNet.bellmanford (models.py:131-141) in one kernel:
    query = relation_representations[arange(bs), r_index]; boundary = zeros with query[b] at row h_index[b];
    with `readout_mlp` also qbias = mlp.0.weight[:, dim:] @ query + mlp.0.bias for the readout (else None).
    materialize=False skips the (bs, num_node, dim) boundary tensor (returned as None): only the gathers run."""
    table = relation_representations.contiguous()
    bs, num_rel, dim = table.shape
    boundary = torch.empty(bs, num_node, dim, dtype=torch.float32, device=table.device) if materialize else None
    query = torch.empty(bs, dim, dtype=torch.float32, device=table.device)
    qbias, w1, b1 = None, None, None
    if readout_mlp is not None and bs <= 1024:
        lin = readout_mlp[0]
        if tuple(lin.weight.shape) == (2 * dim, 2 * dim) and lin.bias is not None and lin.weight.is_contiguous():
            qbias = torch.empty(bs, 2 * dim, dtype=torch.float32, device=table.device)
            w1, b1 = lin.weight.data_ptr(), lin.bias.data_ptr()
    # contiguous copies of strided index views (h_index[:, 0] on the generic path) must stay referenced until the launch
    # is enqueued: a temporary freed inside the argument list hands its memory to the next temporary
    rows = h_index.to(torch.int64).contiguous()
    pick = r_index.to(torch.int64).contiguous()
    check(lib.ultra_query_boundary(boundary.data_ptr() if materialize else None, query.data_ptr(), rows.data_ptr(),
                                   table.data_ptr(), pick.data_ptr(), bs, num_node, num_rel,
                                   dim, w1, b1, qbias.data_ptr() if qbias is not None else None, _stream(table)))
    return boundary, query, qbias


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class RelationProjectionFunction(torch.autograd.Function):
    """out[l] = relu(x W0_l^T + b0_l) W2_l^T + b2_l for every layer's relation_projection MLP (layers.py:80) as ONE autograd
    node over the layers' own parameters: apply(x, w0_0, b0_0, w2_0, b2_0, w0_1, ...) -> one (..., 64) tensor per layer.
    Forward: one launch (ultra_relation_projection_layers); backward: three (csrc/relproj_bwd.hip), the parameter gradients
    returned as slices of stacked buffers.  torch's chain for the same (two batched products over torch.stack-ed parameters)
    was ~ 40 us forward and ~ 330 us backward of a 3.5 ms step at FB15k237's size."""

    @staticmethod
    def forward(ctx, x, *params):
        n_layer = len(params) // 4
        xc = x.contiguous()
        rows = xc.numel() // 64
        w0, b0, w2, b2 = ([p.contiguous() for p in params[k::4]] for k in range(4))
        out = torch.empty((n_layer,) + tuple(xc.shape), dtype=torch.float32, device=xc.device)
        check(lib.ultra_relation_projection_layers(xc.data_ptr(), _ptr_array(w0), _ptr_array(b0), _ptr_array(w2), _ptr_array(b2),
                                                   out.data_ptr(), rows, n_layer, 64, _stream(xc)))
        ctx.save_for_backward(xc, *params)
        return tuple(out.unbind(0))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grad_out):
        xc, *params = ctx.saved_tensors
        n_layer = len(params) // 4
        rows = xc.numel() // 64
        w0, b0, w2 = ([p.contiguous() for p in params[k::4]] for k in range(3))
        gout = [g.contiguous() if g is not None else torch.zeros_like(xc) for g in grad_out]
        dev = xc.device
        gx = torch.empty_like(xc)
        gw0 = torch.empty(n_layer, 64, 64, dtype=torch.float32, device=dev)
        gw2 = torch.empty_like(gw0)
        gb0 = torch.empty(n_layer, 64, dtype=torch.float32, device=dev)
        gb2 = torch.empty_like(gb0)
        nbytes = lib.ultra_relation_projection_backward_workspace(rows, n_layer)
        work = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        check(lib.ultra_relation_projection_backward(xc.data_ptr(), _ptr_array(w0), _ptr_array(b0), _ptr_array(w2), _ptr_array(gout),
                                                     gx.data_ptr(), gw0.data_ptr(), gb0.data_ptr(), gw2.data_ptr(), gb2.data_ptr(),
                                                     work.data_ptr(), nbytes, rows, n_layer, 64, _stream(xc)))
        grads = []
        for l in range(n_layer):
            grads += [gw0[l], gb0[l], gw2[l], gb2[l]]
        return (gx,) + tuple(grads)


def relation_projection_train(x, layers_params):
    """layers_params: [(w0, b0, w2, b2)] per layer (the nn.Linear parameters themselves) -> list of per-layer outputs."""
    flat = [p for group in layers_params for p in group]
    return list(RelationProjectionFunction.apply(x, *flat))


def relation_projection(x, w0, b0, w2, b2):
    """out[l] = relu(x @ w0[l].T + b0[l]) @ w2[l].T + b2[l] for the stacked (n_layer, 64, 64) weights: one MFMA kernel."""
    x = x.contiguous()
    rows = x.numel() // 64
    n_layer = w0.shape[0]
    out = torch.empty((n_layer,) + tuple(x.shape), dtype=torch.float32, device=x.device)
    check(lib.ultra_relation_projection(x.data_ptr(), w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                        out.data_ptr(), rows, n_layer, 64, _stream(x)))
    return out
