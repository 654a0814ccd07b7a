"""Drop-in operator boundary: generalized_rspmm and the six RSPMM*Function classes.

Mirrors /root/reference/ultra/rspmm/rspmm.py (names, argument meaning, error behaviour):
  * generalized_rspmm(edge_index, edge_type, edge_weight, relation, input, sum="add", mul="mul")
    (rspmm.py:168-179) -- accepts unsorted edges, ValueError for unknown (sum, mul) pairs;
  * RSPMM{Add,Min,Max}{Mul,Add}Function (rspmm.py:12-165) -- require sorted `edge_index`
    (AssertionError "Expect sorted `edge_index`"), differentiable w.r.t. edge_weight, relation, input;
  * `rspmm` -- a namespace exporting the reference extension's function names
    rspmm_<sum>_<mul>_{forward,backward}_cuda (rspmm.cpp:270-282), bound to the stateless C entry points.

What changes underneath: the per-call argsort / ind2ptr / host syncs of the reference are replaced by a
cached `Plan` (built once per graph) and the HIP kernels of libultra_amd.so.  CPU tensors raise: this
engine has no CPU path (the reference's `rspmm_*_cpu` names exist only to say so).
"""
import ctypes
import sys
from collections import OrderedDict

import torch
from torch import autograd

from . import _lib
from ._lib import UltraMat, check, lib

module = sys.modules[__name__]

_DTYPES = {torch.float32: _lib.F32, torch.float64: _lib.F64}


_WEIGHT_EPOCH = [0]
# the relation gradient of dense-twin graphs on the matrix cores (A/B switch for tests: the relation-major edge walk is the other side)
DENSE_RELATION_GRAD = True


def tag_edge_weight(edge_weight):
    """Marks a per-step edge-weight vector (a training step's 0/1 keep mask) as unchanged from here on: the plans then bring
    it into their edge order once per step instead of once per rspmm call (ultra_rspmm_weight_epoch).  An in-place write
    to the tensor voids the tag."""
    _WEIGHT_EPOCH[0] += 1
    edge_weight._ultra_epoch = (_WEIGHT_EPOCH[0], edge_weight._version)
    return edge_weight


def _weight_epoch(edge_weight):
    tag = getattr(edge_weight, "_ultra_epoch", None) if edge_weight is not None else None
    return tag[0] if tag is not None and tag[1] == edge_weight._version else 0


def _announce_weight(edge_weight, epoch=None):
    """Tells the library which tagged vector the next weighted call carries (0: untagged)."""
    if edge_weight is not None:
        lib.ultra_rspmm_weight_epoch(int(_weight_epoch(edge_weight) if epoch is None else epoch))


def _stream(t):
    """The current HIP stream of the operand's device (the C entry points make that device current for the launch)."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError("ultra_amd.rspmm: expected a GPU (ROCm `cuda`) tensor, got device `%s`; "
                               "the MI355X engine has no CPU path" % t.device)


def _dtype_code(*tensors):
    dt = tensors[0].dtype
    for t in tensors:
        if t.dtype != dt:
            raise RuntimeError("Expected tensors of the same floating type (edge_weight, relation, input), got %s and %s"
                               % (dt, t.dtype))  # checkAllSameType, rspmm.cpp:23
    if dt not in _DTYPES:
        raise RuntimeError("rspmm supports float32 / float64, got %s" % dt)  # AT_DISPATCH_FLOATING_TYPES
    return _DTYPES[dt]


def as_mat(t):
    """Describe a 2-D (rows, D) or batch-major 3-D (batch, rows, d) tensor to the C ABI without copying."""
    if t.dim() == 2:
        if t.stride(1) != 1 and t.shape[1] > 1:
            t = t.contiguous()
        return t, UltraMat(t.data_ptr(), 1, 0, t.shape[0], t.stride(0), t.shape[1])
    if t.dim() == 3:
        if t.stride(2) != 1 and t.shape[2] > 1:
            t = t.contiguous()
        return t, UltraMat(t.data_ptr(), t.shape[0], t.stride(0), t.shape[1], t.stride(1), t.shape[2])
    raise RuntimeError("Expected a 2-dimensional (or batch-major 3-dimensional) tensor, got %d dims" % t.dim())


class Plan(object):
    """Aggregation plan of one graph (sorted CSR + balanced work list), resident in HBM."""

    TYPE_RUN_MIN_MEAN_LENGTH = 16   # build the type-run twin when (row, type) runs average at least this many edges
    DENSE_MIN_FILL = 0.25           # build the dense-format twin when this share of the (row, type, col) cells holds an edge

    def __init__(self, edge_index, edge_type, num_node, num_relation, seg_len=0, g_max=0, exact_order=False,
                 num_in=None, type_runs="auto", dense="auto"):
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise RuntimeError("Expected `edge_index` of shape (2, num_edge)")          # checkDim/checkSize
        if edge_type.dim() != 1 or edge_type.shape[0] != edge_index.shape[1]:
            raise RuntimeError("Expected `edge_type` of shape (num_edge,)")
        if edge_index.dtype != edge_type.dtype:
            raise RuntimeError("Expected `edge_index` and `edge_type` of the same type")  # checkSameType, rspmm.cpp:22
        ei = edge_index.detach().to("cpu", torch.int64).contiguous()
        et = edge_type.detach().to("cpu", torch.int64).contiguous()
        flags = ((_lib.PLAN_EXACT_ORDER if exact_order else 0) | (_lib.PLAN_TYPE_RUNS if type_runs == "only" else 0)
                 | (_lib.PLAN_DENSE if dense == "only" else 0))
        opts = _lib.PlanOpts(int(seg_len), int(g_max), flags, 0)
        handle = ctypes.c_void_p()
        self.num_edge = ei.shape[1]
        self.num_node = int(num_node)
        self.num_in = int(num_node if num_in is None else num_in)
        self.num_relation = int(num_relation)
        check(lib.ultra_plan_create(ctypes.byref(handle), ei.data_ptr(), et.data_ptr(), self.num_edge, self.num_node,
                                    self.num_in, self.num_relation, ctypes.byref(opts)))
        self._h = handle
        self.exact = bool(exact_order)
        # Dense graphs with few relation types (ULTRA's relation graph: 474 nodes, 4 types, ~470 edges per
        # (row, type) run) get a twin plan whose items hold one relation each; add_mul forwards use it.
        self.typed = None
        self.dense = None
        self._edges = None
        self._dense_t = None
        if dense == "only" or type_runs == "only":
            return
        # (Nearly) complete graphs -- again ULTRA's relation graph -- also get a dense-format twin: fp32 add_mul with unit
        # edge weights then runs on the matrix cores (csrc/rspmm_dense.hip).
        cells = self.num_node * self.num_in * max(self.num_relation, 1)
        if dense in ("auto", True) and self.num_edge > 0 and self.num_in <= _lib.DENSE_MAX_IN_ROW \
                and cells <= (1 << 26) and (dense is True or self.num_edge >= self.DENSE_MIN_FILL * cells):
            try:
                self.dense = Plan(ei, et, num_node, num_relation, num_in=num_in, type_runs=False, dense="only")
                self._edges = (ei, et)      # (a few hundred nodes: kept for the transposed twin of the backward)
            except _lib.UltraError:     # an edge repeated more than 255 times: the edge walk serves it
                if dense is True:
                    raise
            # a reference-order plan only keeps the twin for the reference-order layer kernel (fused_layer), and only
            # when the graph qualifies for it (parallel edges sorted by type, no repeats, at most 4 types)
            if exact_order and self.dense is not None and self.dense.info()["dense_order_bytes"] == 0:
                self.dense = None
        if type_runs in ("auto", True) and not exact_order and self.num_edge > 0:
            runs = max(1, self.info()["n_type_run"])
            if type_runs is True or self.num_edge / runs >= self.TYPE_RUN_MIN_MEAN_LENGTH:
                self.typed = Plan(ei, et, num_node, num_relation, seg_len=seg_len, g_max=g_max, num_in=num_in,
                                  type_runs="only", dense=False)

    def _twin_for(self, sum, mul, edge_weight, input, *others):
        """The specialised twin plan that serves this call, or None for the general (row, col) plan."""
        if sum != "add" or mul != "mul" or self.exact:     # (the twins' kernels re-associate the sum)
            return None
        if self.dense is not None and edge_weight is None and input.dtype == torch.float32 \
                and input.shape[-1] % 32 == 0 and input.dim() in (2, 3) \
                and all(t is None or (t.stride(-1) == 1 and t.data_ptr() % 16 == 0
                                      and all(st % 4 == 0 for st in t.stride()[:-1])) for t in (input,) + others):
            return self.dense
        return self.typed

    def dense_transposed(self):
        """The dense-format plan of the TRANSPOSED graph (built on first use): the input gradient of add_mul,
        input_grad[col] = sum_e rel[type] * output_grad[row] (rspmm.cpp:110-112), is an rspmm forward over it."""
        if self._dense_t is None and self.dense is not None and self._edges is not None:
            ei, et = self._edges
            self._dense_t = Plan(ei.flip(0).contiguous(), et, self.num_in, self.num_relation, num_in=self.num_node,
                                 type_runs=False, dense="only")
        return self._dense_t

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:   # (module globals are gone at interpreter shutdown)
            try:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                lib.ultra_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    def pin(self, delta=1):
        """A captured hipGraph starts (+1) / stops (-1) referencing this plan's device arrays (twins included)."""
        for p in (self, self.typed, self.dense, self._dense_t):
            if p is not None and getattr(p, "_h", None):
                check(lib.ultra_plan_pin(p._h, int(delta)))

    def info(self):
        info = _lib.PlanInfo()
        check(lib.ultra_plan_get_info(self._h, ctypes.byref(info)))
        return {name: getattr(info, name) for name, _ in _lib.PlanInfo._fields_}

    def schedule_info(self, nparts):
        info = _lib.ScheduleInfo()
        check(lib.ultra_plan_schedule_info(self._h, int(nparts), ctypes.byref(info)))
        return {name: getattr(info, name) for name, _ in _lib.ScheduleInfo._fields_ if name != "reserved"}

    def schedule(self, nparts):
        """(chunk_ptr, unit_ptr, units, chunks[n, 4]) of the static schedule for `nparts` workgroups per span."""
        out = []
        for which in range(4):
            n = ctypes.c_int64()
            check(lib.ultra_plan_schedule_export(self._h, int(nparts), which, None, 0, ctypes.byref(n)))
            t = torch.empty(n.value, dtype=torch.int32)
            check(lib.ultra_plan_schedule_export(self._h, int(nparts), which, t.data_ptr(), n.value, ctypes.byref(n)))
            out.append(t.view(-1, 4) if which == 3 else t)
        return tuple(out)

    def streams(self, nparts, walkers=16):
        """(sdesc[nparts * 64, 2] = {first record, steps}, srec[n, 2] = (col, type) records, markers (row, num_relation)).
        walkers=12: the schedule of the launches whose last four waves apply the layer update beside the walk."""
        out = []
        nparts = int(nparts) | ((1 << 24) if walkers == 12 else 0)
        for which in (4, 5):
            n = ctypes.c_int64()
            check(lib.ultra_plan_schedule_export(self._h, nparts, which, None, 0, ctypes.byref(n)))
            t = torch.empty(n.value, dtype=torch.int32)
            check(lib.ultra_plan_schedule_export(self._h, nparts, which, t.data_ptr(), n.value, ctypes.byref(n)))
            out.append(t.view(-1, 2))
        return tuple(out)

    def part_rows(self, nparts):
        """(prow, prow_ptr): the rows each workgroup aggregates -- ascending, -1 padded to whole 32-row tiles -- and their
        bounds per workgroup (schedule arrays 6 and 7: the work list of the update tail)."""
        out = []
        for which in (6, 7):
            n = ctypes.c_int64()
            check(lib.ultra_plan_schedule_export(self._h, int(nparts), which, None, 0, ctypes.byref(n)))
            t = torch.empty(n.value, dtype=torch.int32)
            check(lib.ultra_plan_schedule_export(self._h, int(nparts), which, t.data_ptr(), n.value, ctypes.byref(n)))
            out.append(t)
        return tuple(out)

    def export(self, which):
        n = ctypes.c_int64()
        check(lib.ultra_plan_export(self._h, which, None, 0, ctypes.byref(n)))
        out = torch.empty(n.value, dtype=torch.int32)
        check(lib.ultra_plan_export(self._h, which, out.data_ptr(), n.value, ctypes.byref(n)))
        return out.view(torch.uint8) if which in (_lib.ARR_DENSE, _lib.ARR_DENSE_ORDER) else out

    # ---- kernels ----
    def forward(self, relation, input, edge_weight=None, boundary=None, sum="add", mul="mul", out=None, point=None,
                keep=False, weight_epoch=None):
        """point=(rows, values): a boundary that is zero except row rows[o] of outer slice o, where it is values[o]
        (the NBFNet boundary condition); excludes `boundary`.  sum="add": added to that row only.  sum="min" / "max":
        that row meets values[o], every other row meets 0 (max(update, boundary), layers.py:206-207) -- served by
        reference-order plans; returns None where it is not (the caller then passes the boundary as a tensor).
        keep=True: `edge_weight` is a 0/1 keep mask -- an edge with 0 is absent from the graph for this call
        (ultra_rspmm_forward_masked; differs from a zero weight under min / max only)."""
        if point is not None:
            if boundary is not None:
                raise RuntimeError("a point boundary excludes `boundary`")
            if sum != "add" and not self.exact:
                return None
            twin = self._twin_for(sum, mul, edge_weight, input, relation, point[1], out)
        else:
            twin = self._twin_for(sum, mul, edge_weight, input, relation, boundary, out)
        if twin is not None:
            return twin.forward(relation, input, edge_weight=edge_weight, boundary=boundary, sum=sum, mul=mul, out=out,
                                point=point, keep=keep, weight_epoch=weight_epoch)
        _require_gpu(relation, input, edge_weight, boundary)
        dt = _dtype_code(*([relation, input] + ([edge_weight] if edge_weight is not None else [])
                           + ([boundary] if boundary is not None else [])))
        relation, mrel = as_mat(relation)
        input, mx = as_mat(input)
        if relation.dim() != input.dim():
            raise RuntimeError("relation and input must both be 2-D or both batch-major 3-D")
        if out is None:
            shape = list(input.shape)
            shape[-2] = self.num_node
            out = torch.empty(shape, dtype=input.dtype, device=input.device)
        out, mout = as_mat(out)
        mb = None
        if boundary is not None:
            boundary, mbv = as_mat(boundary)
            mb = ctypes.byref(mbv)
        w = None
        if edge_weight is not None:
            if edge_weight.dim() != 1 or edge_weight.shape[0] != self.num_edge:
                raise RuntimeError("Expected `edge_weight` of shape (num_edge,)")
            if weight_epoch is None:
                weight_epoch = _weight_epoch(edge_weight) if edge_weight.is_contiguous() else 0
            edge_weight = edge_weight.contiguous()
            w = edge_weight.data_ptr()
        if point is not None:
            rows, vals = point
            n_outer = 1 if input.dim() == 2 else input.shape[0]
            rows = rows.to(torch.int64).contiguous()
            vals = vals.reshape(n_outer, 1, vals.shape[-1]) if input.dim() == 3 else vals.reshape(1, vals.shape[-1])
            if rows.numel() != n_outer:
                raise RuntimeError("Expected one boundary row per outer slice (%d), got %d" % (n_outer, rows.numel()))
            _require_gpu(rows, vals)
            vals, mv = as_mat(vals)
            _announce_weight(edge_weight, weight_epoch)
            rc = lib.ultra_rspmm_forward_point(self._h, _lib.SUM_CODES[sum], _lib.MUL_CODES[mul], dt, w, ctypes.byref(mrel),
                                               ctypes.byref(mx), rows.data_ptr(), ctypes.byref(mv), ctypes.byref(mout), _stream(input))
            if rc == _lib.ULTRA_ERR_UNSUPPORTED and sum != "add":
                return None      # (min / max: the caller passes the boundary as a tensor)
            check(rc)
            return out
        entry = lib.ultra_rspmm_forward_masked if (keep and w is not None and sum != "add") else lib.ultra_rspmm_forward
        _announce_weight(edge_weight, weight_epoch)
        check(entry(self._h, _lib.SUM_CODES[sum], _lib.MUL_CODES[mul], dt, w, ctypes.byref(mrel),
                    ctypes.byref(mx), mb, ctypes.byref(mout), _stream(input)))
        return out

    def forward_update(self, relation, input, weight, bias, ln_weight, ln_bias, eps, flags, mul="mul", point=None, timed=None,
                       sum="add"):
        """Aggregate (`sum`, `mul`, optional point boundary) AND the layer update
        `[input +] relu(layer_norm(linear(cat[input, aggregate])))` in one launch (ultra_rspmm_forward_update: the workgroup
        that aggregated a row also updates it).  Bit-equal with forward(point=...) followed by dense.conv_update.
        Returns the layer output, or None where the launch does not serve the call (the caller makes the two calls).
        timed=(warmup, iters): returns (ms per call, ms of the kernel alone) between HIP events instead
        (ultra_rspmm_forward_update_timed)."""
        if not self.exact or input.dtype != torch.float32 or input.dim() != 3 or input.shape[-1] != 64 or not input.is_cuda:
            return None
        _require_gpu(relation, input, weight)
        relation, mrel = as_mat(relation)
        input, mx = as_mat(input)
        agg = torch.empty((input.shape[0], self.num_node, 64), dtype=torch.float32, device=input.device)
        out = torch.empty_like(agg)
        agg, magg = as_mat(agg)
        out, mout = as_mat(out)
        rows_ptr, mv = None, None
        if point is not None:
            rows, vals = point
            rows = rows.to(torch.int64).contiguous()
            if rows.numel() != input.shape[0]:
                raise RuntimeError("Expected one boundary row per outer slice (%d), got %d" % (input.shape[0], rows.numel()))
            vals = vals.reshape(input.shape[0], 1, vals.shape[-1])
            _require_gpu(rows, vals)
            vals, mvv = as_mat(vals)
            rows_ptr, mv = rows.data_ptr(), ctypes.byref(mvv)
        weight = weight.contiguous()
        args = (self._h, _lib.SUM_CODES[sum], _lib.MUL_CODES[mul], ctypes.byref(mrel), ctypes.byref(mx), rows_ptr, mv, ctypes.byref(magg),
                weight.data_ptr(),
                _ptr(bias), _ptr(ln_weight), _ptr(ln_bias), float(eps), int(flags), ctypes.byref(mout), _stream(input))
        if timed is not None:
            ms, ms_kernel = ctypes.c_float(), ctypes.c_float()
            rc = lib.ultra_rspmm_forward_update_timed(*(args + (int(timed[0]), int(timed[1]), ctypes.byref(ms), ctypes.byref(ms_kernel))))
            if rc == _lib.ULTRA_ERR_UNSUPPORTED:
                return None
            check(rc)
            return ms.value, ms_kernel.value
        rc = lib.ultra_rspmm_forward_update(*args)
        if rc == _lib.ULTRA_ERR_UNSUPPORTED:
            return None
        check(rc)
        return out

    def forward_onehot(self, relation, input, src_rows, edge_weight=None, boundary=None):
        """add_mul forward for an input that is zero outside row src_rows[o] of every outer slice (the NBFNet
        layer-0 boundary condition).  Same result as forward(sum="add", mul="mul"), visiting only the edges
        gathered from the source rows."""
        _require_gpu(relation, input, edge_weight, boundary, src_rows)
        dt = _dtype_code(*([relation, input] + ([edge_weight] if edge_weight is not None else [])
                           + ([boundary] if boundary is not None else [])))
        relation, mrel = as_mat(relation)
        input, mx = as_mat(input)
        shape = list(input.shape)
        shape[-2] = self.num_node
        out = torch.empty(shape, dtype=input.dtype, device=input.device)
        out, mout = as_mat(out)
        n_outer = 1 if input.dim() == 2 else input.shape[0]
        src_rows = src_rows.to(torch.int64).contiguous()
        if src_rows.numel() != n_outer:
            raise RuntimeError("Expected one source row per outer slice (%d), got %d" % (n_outer, src_rows.numel()))
        mb = None
        if boundary is not None:
            boundary, mbv = as_mat(boundary)
            mb = ctypes.byref(mbv)
        w = None
        if edge_weight is not None:
            edge_weight = edge_weight.contiguous()
            w = edge_weight.data_ptr()
        check(lib.ultra_rspmm_forward_onehot(self._h, dt, w, ctypes.byref(mrel), ctypes.byref(mx), src_rows.data_ptr(), mb,
                                             ctypes.byref(mout), _stream(input)))
        return out

    def fused_layer(self, relation, input, linear, layer_norm=None, relu=True, residual=False, boundary=None, point=None):
        """A whole layer -- add_mul aggregate (+ boundary), Linear(cat[input, agg]), LayerNorm, ReLU, residual -- in one
        launch on the dense-format twin (ultra_nbf_dense_layer).  Returns None when this plan / these operands are not
        served (the caller then runs forward() + the update kernel)."""
        d = self.dense
        if d is None or self.num_relation > 4 or self.num_node != self.num_in or input.dim() != 3 or input.shape[-1] != 64 \
                or input.dtype != torch.float32 or relation.dtype != torch.float32 or tuple(linear.weight.shape) != (64, 128):
            return None
        others = (relation, boundary, point[1] if point is not None else None)
        if not all(t is None or (t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and all(st % 4 == 0 for st in t.stride()[:-1]))
                   for t in (input,) + others):
            return None
        relation, mrel = as_mat(relation)
        input, mx = as_mat(input)
        out = torch.empty_like(input)
        _, mout = as_mat(out)
        mb, rows_ptr = None, None
        if point is not None:
            rows, vals = point
            rows = rows.to(torch.int64).contiguous()
            vals, mbv = as_mat(vals.reshape(input.shape[0], 1, 64))
            mb, rows_ptr = ctypes.byref(mbv), rows.data_ptr()
        elif boundary is not None:
            boundary, mbv = as_mat(boundary)
            mb = ctypes.byref(mbv)
        flags = (1 if layer_norm is not None else 0) | (2 if relu else 0) | (4 if residual else 0) \
            | (_lib.LAYER_REFERENCE_ORDER if self.exact else 0)
        check(lib.ultra_nbf_dense_layer(d._h, ctypes.byref(mrel), ctypes.byref(mx), mb, rows_ptr, linear.weight.data_ptr(),
                                        linear.bias.data_ptr() if linear.bias is not None else None,
                                        layer_norm.weight.data_ptr() if layer_norm is not None else None,
                                        layer_norm.bias.data_ptr() if layer_norm is not None else None,
                                        float(layer_norm.eps) if layer_norm is not None else 1e-5, flags, ctypes.byref(mout),
                                        _stream(input)))
        return out

    def layer0_fill(self, batch_size, linear, layer_norm=None, relu=True, device=None):
        """The constant rows of layer0() -- relu(LayerNorm(bias)) everywhere -- into a fresh (batch, N, 64) tensor: they depend
        on the layer's parameters only, so a caller may launch this early (on a side stream, beside the relation model) and
        hand the tensor to layer0(out=...) for the special rows."""
        device = device if device is not None else linear.weight.device
        out = torch.empty(batch_size, self.num_node, 64, dtype=torch.float32, device=device)
        _, mout = as_mat(out)
        flags = (1 if layer_norm is not None else 0) | (2 if relu else 0) | 16
        check(lib.ultra_nbf_layer0(self._h, None, None, None, None, linear.weight.data_ptr(),
                                   linear.bias.data_ptr() if linear.bias is not None else None,
                                   layer_norm.weight.data_ptr() if layer_norm is not None else None,
                                   layer_norm.bias.data_ptr() if layer_norm is not None else None,
                                   float(layer_norm.eps) if layer_norm is not None else 1e-5, flags, ctypes.byref(mout),
                                   _stream(out)))
        return out

    def layer0(self, relation, src_rows, src_values, linear, layer_norm=None, relu=True, residual=False, edge_weight=None,
               aggregate="sum", out=None):
        """Layer 0 of an NBFNet on its one-hot boundary condition (ultra_nbf_layer0): returns the (batch, N, 64) hidden
        state of `relu(LayerNorm(linear(cat[x0, rspmm(x0) + x0]))) [+ x0]`, x0 = src_values[b] (ones if None) at row
        src_rows[b] and zero elsewhere, without materialising x0 or the aggregate."""
        _require_gpu(relation, src_rows, src_values, edge_weight)
        relation, mrel = as_mat(relation)
        bs = relation.shape[0]
        prefilled = out is not None       # (layer0_fill wrote the constant rows already)
        if out is None:
            out = torch.empty(bs, self.num_node, 64, dtype=torch.float32, device=relation.device)
        elif tuple(out.shape) != (bs, self.num_node, 64) or out.dtype != torch.float32 or not out.is_contiguous():
            raise RuntimeError("layer0(out=...): expected the (batch, num_node, 64) fp32 tensor of layer0_fill")
        _, mout = as_mat(out)
        src_rows = src_rows.to(torch.int64).contiguous()
        if src_values is not None:
            src_values = src_values.contiguous()
        if edge_weight is not None:
            edge_weight = edge_weight.to(torch.float32).contiguous()
        flags = (1 if layer_norm is not None else 0) | (2 if relu else 0) | (4 if residual else 0) | (8 if aggregate == "max" else 0) \
            | (32 if prefilled else 0)
        check(lib.ultra_nbf_layer0(self._h, edge_weight.data_ptr() if edge_weight is not None else None, ctypes.byref(mrel),
                                   src_rows.data_ptr(), src_values.data_ptr() if src_values is not None else None,
                                   linear.weight.data_ptr(), linear.bias.data_ptr() if linear.bias is not None else None,
                                   layer_norm.weight.data_ptr() if layer_norm is not None else None,
                                   layer_norm.bias.data_ptr() if layer_norm is not None else None,
                                   float(layer_norm.eps) if layer_norm is not None else 1e-5, flags, ctypes.byref(mout),
                                   _stream(relation)))
        return out

    def backward(self, relation, input, output, output_grad, edge_weight=None, need_weight_grad=False, sum="add",
                 mul="mul", weight_epoch=None, input_grad_base=None):
        """input_grad_base (sum == "add"): a tensor of the input's shape that the returned input gradient starts from (the
        input's gradient from another consumer); it is overwritten with the total and returned."""
        _require_gpu(relation, input, output, output_grad, edge_weight)
        dt = _dtype_code(relation, input, output, output_grad)
        relation, mrel = as_mat(relation)
        input, mx = as_mat(input)
        output, mo = as_mat(output)
        output_grad, mog = as_mat(output_grad)
        rgrad = torch.empty(relation.shape, dtype=relation.dtype, device=relation.device)
        _, mrg = as_mat(rgrad)
        # A graph with a dense-format twin (ULTRA's relation graph) takes its input gradient as that twin's forward over
        # the transposed graph -- the matrix-core kernel of the forward pass, 15 us where the edge walk over the
        # transposed plan takes 70 -- and asks ultra_rspmm_backward for the relation gradient alone.
        xgrad, mxg = None, None
        if self._twin_for(sum, mul, edge_weight, output_grad, relation) is self.dense and self.dense is not None \
                and input.dtype == torch.float32:
            twin = self.dense_transposed()
            if twin is not None:
                xgrad = twin.forward(relation, output_grad, boundary=input_grad_base)
            # ... and its relation gradient from the same format: the per-type products A_t . x of the forward kernel, weighed
            # with output_grad (ultra_rspmm_dense_relation_grad) -- 15 us where the walk over the relation-major plan took 71 + 12
            if xgrad is not None and not need_weight_grad and DENSE_RELATION_GRAD:
                rc = lib.ultra_rspmm_dense_relation_grad(self.dense._h, ctypes.byref(mx), ctypes.byref(mog), ctypes.byref(mrg),
                                                         _stream(input))
                if rc == _lib.ULTRA_OK:
                    return None, rgrad, xgrad
                if rc != _lib.ULTRA_ERR_UNSUPPORTED:
                    check(rc)
        base = None
        if xgrad is None:
            if input_grad_base is not None:
                if sum != "add" or tuple(input_grad_base.shape) != tuple(input.shape) or input_grad_base.dtype != input.dtype:
                    raise RuntimeError("input_grad_base: the input's shape and dtype, sum == 'add'")
                xgrad = input_grad_base if input_grad_base.is_contiguous() else input_grad_base.contiguous()
                base = xgrad
            else:
                xgrad = torch.empty(input.shape, dtype=input.dtype, device=input.device)
            _, mxg = as_mat(xgrad)
        w = None
        if edge_weight is not None:
            if weight_epoch is None:
                weight_epoch = _weight_epoch(edge_weight) if edge_weight.is_contiguous() else 0
            edge_weight = edge_weight.contiguous()
            w = edge_weight.data_ptr()
        wgrad = None
        wg = None
        if need_weight_grad:
            wgrad = torch.zeros(self.num_edge, dtype=input.dtype, device=input.device)
            wg = wgrad.data_ptr()
        _announce_weight(edge_weight, weight_epoch)
        if base is not None:      # (in place: every row's base is read by the thread that writes its total)
            check(lib.ultra_rspmm_backward_add(self._h, _lib.SUM_CODES[sum], _lib.MUL_CODES[mul], dt, w, ctypes.byref(mrel),
                                               ctypes.byref(mx), ctypes.byref(mo), ctypes.byref(mog), wg, ctypes.byref(mrg),
                                               ctypes.byref(mxg), ctypes.byref(mxg), _stream(input)))
        else:
            check(lib.ultra_rspmm_backward(self._h, _lib.SUM_CODES[sum], _lib.MUL_CODES[mul], dt, w, ctypes.byref(mrel),
                                           ctypes.byref(mx), ctypes.byref(mo), ctypes.byref(mog), wg, ctypes.byref(mrg),
                                           ctypes.byref(mxg) if mxg is not None else None, _stream(input)))
        return wgrad, rgrad, xgrad

    def forward_timed(self, relation, input, edge_weight=None, boundary=None, sum="add", mul="mul", warmup=3, iters=20,
                      point=None):
        """Mean HIP-event time (ms) of the forward launch sequence on the current stream; `point` as in forward()."""
        twin = self._twin_for(sum, mul, edge_weight, input, relation, boundary if point is None else point[1])
        if twin is not None:
            res = twin.forward_timed(relation, input, edge_weight=edge_weight, boundary=boundary, sum=sum,
                                     mul=mul, warmup=warmup, iters=iters, point=point)
            self.last_main_kernel_ms = twin.last_main_kernel_ms
            return res
        dt = _dtype_code(relation, input)
        relation, mrel = as_mat(relation)
        input, mx = as_mat(input)
        shape = list(input.shape)
        shape[-2] = self.num_node
        out = torch.empty(shape, dtype=input.dtype, device=input.device)
        out, mout = as_mat(out)
        mb, rows_ptr = None, None
        if point is not None:
            rows, vals = point
            rows = rows.to(torch.int64).contiguous()
            n_outer = 1 if input.dim() == 2 else input.shape[0]
            vals, mbv = as_mat(vals.reshape(n_outer, 1, vals.shape[-1]) if input.dim() == 3 else vals.reshape(1, vals.shape[-1]))
            mb, rows_ptr = ctypes.byref(mbv), rows.data_ptr()
        elif boundary is not None:
            boundary, mbv = as_mat(boundary)
            mb = ctypes.byref(mbv)
        edge_weight = edge_weight.contiguous() if edge_weight is not None else None
        w = edge_weight.data_ptr() if edge_weight is not None else None
        ms, ms_kernel = ctypes.c_float(), ctypes.c_float()
        check(lib.ultra_rspmm_forward_timed(self._h, _lib.SUM_CODES[sum], _lib.MUL_CODES[mul], dt, w, ctypes.byref(mrel),
                                            ctypes.byref(mx), mb, rows_ptr, ctypes.byref(mout), _stream(input), warmup, iters,
                                            ctypes.byref(ms), ctypes.byref(ms_kernel)))
        self.last_main_kernel_ms = ms_kernel.value
        return ms.value, out


# ---- plan cache: the graph is static across the 12 rspmm calls of a forward and across batches ----
_PLAN_CACHE = OrderedDict()
_PLAN_CACHE_SIZE = 16
# Plans built for the operator / module API sum in the reference's order (rspmm.cpp:61-72) by default: scores and
# rankings then reproduce the reference's.  exact_order=False selects the re-associating plans (split hub rows,
# type-run / dense-format twins): same sums up to fp32 rounding, a different rounding pattern.
_plan_defaults = {"seg_len": 0, "g_max": 0, "exact_order": True, "type_runs": "auto", "dense": "auto"}


def set_plan_defaults(seg_len=0, g_max=0, exact_order=True, type_runs="auto", dense="auto"):
    """Tuning hook: defaults for newly built plans (clears the cache)."""
    _plan_defaults.update(seg_len=seg_len, g_max=g_max, exact_order=exact_order, type_runs=type_runs, dense=dense)
    _PLAN_CACHE.clear()


def get_plan(edge_index, edge_type, num_node, num_relation, exact_order=None):
    """The cached plan of a graph.  exact_order=None: the default kind (set_plan_defaults; reference summation order);
    False: the re-associating kind, whatever the default -- what the models' training step asks for (its forward feeds a
    scatter-add backward and a stochastic optimiser step: no summation order to reproduce there)."""
    defaults = _plan_defaults if exact_order is None else dict(_plan_defaults, exact_order=bool(exact_order))
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), tuple(edge_index.stride()),
           edge_type.data_ptr(), edge_type._version, tuple(edge_type.shape), str(edge_index.device),
           int(num_node), int(num_relation), defaults["exact_order"])
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        plan, ei_ref, et_ref = hit
        _PLAN_CACHE.move_to_end(key)
        if _PLAN_RECORDER is not None:
            _PLAN_RECORDER.append(plan)
        return plan
    plan = Plan(edge_index, edge_type, num_node, num_relation, **defaults)
    if _PLAN_RECORDER is not None:
        _PLAN_RECORDER.append(plan)
    # the tensors are kept alive with the plan so a recycled data_ptr can never alias a stale entry
    _PLAN_CACHE[key] = (plan, edge_index, edge_type)
    while len(_PLAN_CACHE) > _PLAN_CACHE_SIZE:
        _PLAN_CACHE.popitem(last=False)
    return plan


def clear_plan_cache():
    _PLAN_CACHE.clear()


def cached_plans():
    """The Plan objects currently held by the cache."""
    return [entry[0] for entry in _PLAN_CACHE.values()]


_PLAN_RECORDER = None


class record_plans(object):
    """with record_plans() as used: ...   -- `used` collects the plans get_plan() hands out inside the block (each once).
    A hipGraph capture pins exactly the plans its warm-up runs asked for (graph.py), not whatever else sits in the cache."""

    def __enter__(self):
        global _PLAN_RECORDER
        self._outer = _PLAN_RECORDER
        self._raw = []
        _PLAN_RECORDER = self._raw
        self.plans = []
        return self

    def __exit__(self, *exc):
        global _PLAN_RECORDER
        _PLAN_RECORDER = self._outer
        seen = set()
        for plan in self._raw:
            if id(plan) not in seen:
                seen.add(id(plan))
                self.plans.append(plan)
        if self._outer is not None:
            self._outer.extend(self.plans)
        return False


class _PlanRSPMM(autograd.Function):
    """autograd node shared by every (sum, mul) pair; the graph plan rides along as a non-tensor arg."""

    @staticmethod
    def forward(ctx, plan, sum, mul, edge_weight, relation, input, boundary=None, keep=False, point_rows=None,
                point_values=None):
        """boundary (sum == "add" only): added in the kernel's epilogue (layers.py:199-200), its gradient is the output
        gradient itself.  keep: `edge_weight` is a 0/1 keep mask (Plan.forward).  point_rows / point_values (sum == "add",
        excludes `boundary`): the boundary condition in closed form -- point_values[b] at row point_rows[b] of sample b,
        zero elsewhere; its gradient is those rows of the output gradient, so the (batch, N, d) gradient of a boundary
        TENSOR -- which six layers would each hand to autograd to be summed -- never exists."""
        if (boundary is not None or point_rows is not None) and sum != "add":
            raise RuntimeError("the fused boundary of the differentiable rspmm serves the sum aggregate only")
        point = (point_rows, point_values) if point_rows is not None else None
        output = plan.forward(relation, input, edge_weight=edge_weight, boundary=boundary, sum=sum, mul=mul, keep=keep,
                              point=point)
        ctx.plan, ctx.sum, ctx.mul = plan, sum, mul
        ctx.point_rows = point_rows
        ctx.weight_epoch = _weight_epoch(edge_weight)       # (the tag rides on the Python object: read it while it is at hand)
        ctx.save_for_backward(edge_weight, relation, input, output)   # rspmm.py:25
        return output

    @staticmethod
    def backward(ctx, output_grad):
        edge_weight, relation, input, output = ctx.saved_tensors
        need_w = ctx.needs_input_grad[3]
        output_grad = output_grad.contiguous()
        weight_grad, relation_grad, input_grad = ctx.plan.backward(
            relation, input, output, output_grad, edge_weight=edge_weight, need_weight_grad=need_w,
            sum=ctx.sum, mul=ctx.mul,
            weight_epoch=ctx.weight_epoch if (edge_weight is not None and edge_weight.is_contiguous()) else 0)
        boundary_grad = output_grad if ctx.needs_input_grad[6] else None
        values_grad = None
        if ctx.point_rows is not None and ctx.needs_input_grad[9]:
            rows = ctx.point_rows
            values_grad = (output_grad.gather(1, rows.view(-1, 1, 1).expand(-1, 1, output_grad.shape[-1])).squeeze(1) if output_grad.dim() == 3
                           else output_grad[rows[0]].unsqueeze(0))
        return None, None, None, weight_grad, relation_grad, input_grad, boundary_grad, None, None, values_grad   # rspmm.py:35


_OUT_CSR_CACHE = OrderedDict()


def out_edge_csr(edge_index, edge_type, num_node):
    """Edges grouped by their SOURCE node (edge_index[1], the gathered side), every node's edges sorted by type:
    (ptr (num_node + 1), edge ids in (source, type) order, largest out-degree); built once per edge list and kept with it."""
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), edge_type.data_ptr(), edge_type._version,
           int(num_node))
    hit = _OUT_CSR_CACHE.get(key)
    if hit is None:
        src = edge_index[1]
        count = torch.bincount(src, minlength=int(num_node))
        ptr = torch.zeros(int(num_node) + 1, dtype=torch.int64, device=src.device)
        ptr[1:] = count.cumsum(0)
        num_type = int(edge_type.max()) + 1 if edge_type.numel() else 1
        order = torch.sort(src * num_type + edge_type, stable=True)[1].contiguous()
        hit = (ptr, order, int(count.max()) if count.numel() else 0, edge_index, edge_type)      # (the tensors stay with their entry)
        _OUT_CSR_CACHE[key] = hit
        while len(_OUT_CSR_CACHE) > _PLAN_CACHE_SIZE:
            _OUT_CSR_CACHE.popitem(last=False)
    return hit[:3]


class _OnehotRSPMM(autograd.Function):
    """Differentiable add_mul rspmm of an NBFNet's FIRST layer: the input is the boundary condition itself -- values[b]
    at row rows[b] of sample b, zero elsewhere (models.py:59-66, 135-141) -- and so is the boundary added to the sum
    (layers.py:199-200).  Only the edges leaving the source rows carry a message:

        out[b, row_e] += w_e rel[b, type_e] * values[b]        for e with col_e == rows[b];     out[b, rows[b]] += values[b]

    Forward: Plan.forward_onehot (those edges only).  Backward: the same few edges -- S[b, t] = sum of w_e
    output_grad[b, row_e] over the source's out-edges of type t gives relation_grad[b, t] = values[b] * S[b, t] and
    values_grad[b] = sum_t rel[b, t] * S[b, t] + output_grad[b, rows[b]] -- as a handful of small torch kernels over a
    (batch, largest out-degree) padded edge table, instead of two walks over every edge of the graph (at YAGO3-10's
    size 0.56 + 0.89 ms of a 21 ms step, plus 0.56 for the full forward walk).  `dense_input` is the boundary as a
    tensor, which the layer's update reads anyway; its gradient is returned through `values`."""

    @staticmethod
    def forward(ctx, plan, edge_index, edge_type, edge_weight, relation, rows, values, dense_input):
        out = plan.forward_onehot(relation, dense_input, rows, edge_weight=edge_weight, boundary=dense_input)
        ctx.plan, ctx.edge_index, ctx.edge_type = plan, edge_index, edge_type
        ctx.save_for_backward(edge_weight, relation, rows, values)
        return out

    @staticmethod
    def backward(ctx, output_grad):
        edge_weight, relation, rows, values = ctx.saved_tensors
        edge_index, edge_type = ctx.edge_index, ctx.edge_type
        og = output_grad.contiguous()
        need_rel, need_val = ctx.needs_input_grad[4], ctx.needs_input_grad[6]
        ptr, order, max_deg = out_edge_csr(edge_index, edge_type, ctx.plan.num_in)
        grads = _onehot_backward_kernel(ptr, order, edge_index, edge_type, edge_weight, relation, rows, values, og,
                                        need_rel, need_val)
        if grads is None:       # (shapes the kernel does not serve)
            grads = _onehot_backward_torch(ptr, order, max_deg, edge_index, edge_type, edge_weight, relation, rows, values, og,
                                           need_rel, need_val)
        return None, None, None, None, grads[0], None, grads[1], None


def _onehot_backward_kernel(ptr, order, edge_index, edge_type, edge_weight, relation, rows, values, og, need_rel, need_val):
    """(relation_grad, values_grad) of _OnehotRSPMM in one launch (csrc/onehot_bwd.hip); None where it does not apply."""
    if not (og.is_cuda and og.dtype == torch.float32 and relation.dtype == torch.float32 and values.dtype == torch.float32
            and og.dim() == 3 and edge_index.dtype == torch.int64 and edge_type.dtype == torch.int64
            and (edge_weight is None or edge_weight.dtype == torch.float32)):
        return None
    relation, mrel = as_mat(relation)
    og, mog = as_mat(og)
    values = values.contiguous()
    rows = rows.to(torch.int64).contiguous()
    target = edge_index[0].contiguous()
    weight = edge_weight.contiguous() if edge_weight is not None else None
    bs, dim = og.shape[0], og.shape[2]
    rel_grad = torch.empty(bs, relation.shape[1], dim, dtype=torch.float32, device=og.device) if need_rel else None
    val_grad = torch.empty(bs, dim, dtype=torch.float32, device=og.device) if need_val else None
    rc = lib.ultra_rspmm_onehot_backward(ptr.data_ptr(), order.data_ptr(), target.data_ptr(), edge_type.data_ptr(),
                                         weight.data_ptr() if weight is not None else None, ctypes.byref(mrel),
                                         values.data_ptr(), rows.data_ptr(), ctypes.byref(mog),
                                         rel_grad.data_ptr() if need_rel else None, val_grad.data_ptr() if need_val else None,
                                         _stream(og))
    if rc == _lib.ULTRA_ERR_UNSUPPORTED:
        return None
    check(rc)
    return rel_grad, val_grad


def _onehot_backward_torch(ptr, order, max_deg, edge_index, edge_type, edge_weight, relation, rows, values, og, need_rel, need_val):
    """The same gradients as a handful of torch kernels over a (batch, largest out-degree) padded edge table: the
    restatement the kernel is tested against, and the route for shapes it does not serve."""
    bs, _, dim = og.shape
    num_rel = relation.shape[1]
    dev = og.device
    batch_ids = torch.arange(bs, device=dev)
    own = og[batch_ids, rows]                                            # (bs, dim): the boundary's share
    if max_deg == 0 or edge_index.shape[1] == 0:
        return (torch.zeros(bs, num_rel, dim, dtype=og.dtype, device=dev) if need_rel else None), (own if need_val else None)
    start = ptr[rows]
    deg = ptr[rows + 1] - start
    slot = torch.arange(max_deg, device=dev).unsqueeze(0)                # (1, max_deg)
    valid = slot < deg.unsqueeze(1)                                      # (bs, max_deg)
    edge = order[(start.unsqueeze(1) + slot).clamp_(max=edge_index.shape[1] - 1)]
    weight = valid.to(og.dtype)
    if edge_weight is not None:
        weight = weight * edge_weight[edge].to(og.dtype)
    picked = og.gather(1, edge_index[0][edge].unsqueeze(-1).expand(-1, -1, dim)) * weight.unsqueeze(-1)
    cell = (edge_type[edge] + num_rel * batch_ids.unsqueeze(1)).flatten()
    s = og.new_zeros(bs * num_rel, dim).index_add_(0, cell, picked.flatten(0, 1)).view(bs, num_rel, dim)
    return (values.unsqueeze(1) * s if need_rel else None), ((relation * s).sum(dim=1) + own if need_val else None)


def onehot_rspmm(plan, edge_index, edge_type, relation, rows, values, dense_input, edge_weight=None):
    """Differentiable first-layer rspmm on the boundary condition (rows, values); see _OnehotRSPMM."""
    return _OnehotRSPMM.apply(plan, edge_index, edge_type, edge_weight, relation, rows, values, dense_input)


def _check_args(edge_index, edge_type, edge_weight, relation, input):
    # rspmm_forward_check, rspmm.cpp:15-27
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise RuntimeError("Expected 2-dimensional `edge_index` of size (2, num_edge)")
    if edge_type.dim() != 1 or edge_weight.dim() != 1 or relation.dim() != 2 or input.dim() != 2:
        raise RuntimeError("Expected 1-dimensional edge_type / edge_weight and 2-dimensional relation / input")
    E = edge_index.shape[1]
    if edge_type.shape[0] != E or edge_weight.shape[0] != E:
        raise RuntimeError("Expected edge_type and edge_weight of size (%d,)" % E)
    if relation.shape[1] != input.shape[1]:
        raise RuntimeError("Expected relation.size(1) == input.size(1), got %d and %d"
                           % (relation.shape[1], input.shape[1]))
    _require_gpu(edge_index, edge_type, edge_weight, relation, input)
    _dtype_code(edge_weight, relation, input)


def _sorted_assert(edge_index):
    node_in, node_out = edge_index
    if node_in.numel():
        key = node_in * (node_out.max() + 1) + node_out
        assert (key.diff() >= 0).all(), "Expect sorted `edge_index`"   # rspmm.py:16-18


def _make_function(sum, mul):
    class _Function(autograd.Function):
        @staticmethod
        def forward(ctx, edge_index, edge_type, edge_weight, relation, input):
            _check_args(edge_index, edge_type, edge_weight, relation, input)
            _sorted_assert(edge_index)
            plan = get_plan(edge_index, edge_type, input.shape[0], relation.shape[0])
            output = plan.forward(relation, input, edge_weight=edge_weight, sum=sum, mul=mul)
            ctx.plan = plan
            ctx.save_for_backward(edge_index, edge_type, edge_weight, relation, input, output)
            return output

        @staticmethod
        def backward(ctx, output_grad):
            edge_index, edge_type, edge_weight, relation, input, output = ctx.saved_tensors
            weight_grad, relation_grad, input_grad = ctx.plan.backward(
                relation, input, output, output_grad.contiguous(), edge_weight=edge_weight,
                need_weight_grad=True, sum=sum, mul=mul)
            return None, None, weight_grad, relation_grad, input_grad

    _Function.__name__ = _Function.__qualname__ = "RSPMM%s%sFunction" % (sum.capitalize(), mul.capitalize())
    return _Function


RSPMMAddMulFunction = _make_function("add", "mul")   # rspmm.py:12
RSPMMMinMulFunction = _make_function("min", "mul")   # rspmm.py:38
RSPMMMaxMulFunction = _make_function("max", "mul")   # rspmm.py:64
RSPMMAddAddFunction = _make_function("add", "add")   # rspmm.py:90
RSPMMMinAddFunction = _make_function("min", "add")   # rspmm.py:116
RSPMMMaxAddFunction = _make_function("max", "add")   # rspmm.py:142


def generalized_rspmm(edge_index, edge_type, edge_weight, relation, input, sum="add", mul="mul"):
    """rspmm.py:168-179.  Unsorted edges are fine: the cached plan carries the sort."""
    name = "RSPMM%s%sFunction" % (sum.capitalize(), mul.capitalize())
    if not hasattr(module, name):
        raise ValueError("No generalized rspmm implementation found for summation `%s` and multiplication `%s`"
                         % (sum, mul))
    _check_args(edge_index, edge_type, edge_weight, relation, input)
    plan = get_plan(edge_index, edge_type, input.shape[0], relation.shape[0])
    return _PlanRSPMM.apply(plan, sum, mul, edge_weight, relation, input, None, False)


def plan_rspmm(plan, relation, input, edge_weight=None, sum="add", mul="mul", boundary=None, keep=False, point=None):
    """Differentiable rspmm on an explicit plan; accepts batch-major (batch, N, d) operands.  point=(rows, values): the
    boundary condition in closed form (sum == "add"; excludes `boundary`), differentiable in `values`."""
    if point is not None:
        if boundary is not None:
            raise RuntimeError("a point boundary excludes `boundary`")
        return _PlanRSPMM.apply(plan, sum, mul, edge_weight, relation, input, None, keep, point[0], point[1])
    return _PlanRSPMM.apply(plan, sum, mul, edge_weight, relation, input, boundary, keep)


class _ReferenceExports(object):
    """The reference extension's pybind surface (rspmm.cpp:256-283) over the stateless C entry points."""

    def __getattr__(self, name):
        parts = name.split("_")
        if len(parts) == 5 and parts[0] == "rspmm" and parts[1] in _lib.SUM_CODES and parts[2] in _lib.MUL_CODES \
                and parts[3] in ("forward", "backward") and parts[4] in ("cuda", "cpu"):
            if parts[4] == "cpu":
                def no_cpu(*args, **kwargs):
                    raise RuntimeError("ultra_amd: `%s` -- this engine is MI355X-only and has no CPU path" % name)
                return no_cpu
            return self._forward(parts[1], parts[2]) if parts[3] == "forward" else self._backward(parts[1], parts[2])
        raise AttributeError(name)

    @staticmethod
    def _forward(sum, mul):
        fn = getattr(lib, "ultra_rspmm_%s_%s_forward_cuda" % (sum, mul))

        def forward(edge_index, edge_type, edge_weight, relation, input):
            _check_args(edge_index, edge_type, edge_weight, relation, input)
            dt = _dtype_code(edge_weight, relation, input)
            ei, et, ew = edge_index.contiguous(), edge_type.contiguous(), edge_weight.contiguous()
            rel, x = relation.contiguous(), input.contiguous()
            out = torch.empty_like(x)
            check(fn(ei.data_ptr(), et.data_ptr(), ew.data_ptr(), rel.data_ptr(), x.data_ptr(), out.data_ptr(),
                     ei.shape[1], x.shape[0], rel.shape[0], x.shape[1], dt, _stream(input)))
            return out
        return forward

    @staticmethod
    def _backward(sum, mul):
        fn = getattr(lib, "ultra_rspmm_%s_%s_backward_cuda" % (sum, mul))

        def backward(edge_index, edge_type, edge_weight, relation, input, output, output_grad):
            _check_args(edge_index, edge_type, edge_weight, relation, input)
            dt = _dtype_code(edge_weight, relation, input, output, output_grad)
            ei, et, ew = edge_index.contiguous(), edge_type.contiguous(), edge_weight.contiguous()
            rel, x = relation.contiguous(), input.contiguous()
            o, og = output.contiguous(), output_grad.contiguous()
            wg, rg, xg = torch.zeros_like(ew), torch.zeros_like(rel), torch.zeros_like(x)
            check(fn(ei.data_ptr(), et.data_ptr(), ew.data_ptr(), rel.data_ptr(), x.data_ptr(), o.data_ptr(),
                     og.data_ptr(), wg.data_ptr(), rg.data_ptr(), xg.data_ptr(), ei.shape[1], x.shape[0],
                     rel.shape[0], x.shape[1], dt, _stream(input)))
            return wg, rg, xg
        return backward


rspmm = _ReferenceExports()


def get_tuning():
    """The launch knobs in force (ultra_get_tuning) as a dict."""
    t = _lib.Tuning()
    check(lib.ultra_get_tuning(ctypes.byref(t)))
    return {"threads": t.threads, "grid": t.grid, "rel_lds": t.rel_lds, "x_lds": t.x_lds, "unroll": t.unroll,
            "general_walk": t.reserved[0], "unit_walk": t.reserved[1], "update_form": t.reserved[2]}


class tuning_scope(object):
    """with tuning_scope(grid=192): ... -- the launches inside run (and a capture inside records them) with the given knobs changed,
    every other knob as it was; the former tuning is back afterwards.  The knobs are process-wide (ultra_set_tuning): not for
    threads that launch at the same time with different ones."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        self.was = _lib.Tuning()
        check(lib.ultra_get_tuning(ctypes.byref(self.was)))
        t = _lib.Tuning()
        ctypes.memmove(ctypes.byref(t), ctypes.byref(self.was), ctypes.sizeof(t))
        for k, v in self.knobs.items():
            if k in ("general_walk", "unit_walk", "update_form"):
                t.reserved[("general_walk", "unit_walk", "update_form").index(k)] = int(v)
            else:
                setattr(t, k, int(v))
        check(lib.ultra_set_tuning(ctypes.byref(t)))
        return self

    def __exit__(self, *exc):
        check(lib.ultra_set_tuning(ctypes.byref(self.was)))
        return False


def check_device_error():
    """Raise if a launch before this point ended on a bounded wait of the one-launch layer's hand-off (ultra_device_error,
    include/ultra_rspmm.h): call after synchronising.  The same word is looked at on entry of every rspmm forward."""
    check(lib.ultra_device_error())


def set_tuning(threads=0, grid=0, rel_lds=-1, x_lds=-1, unroll=0, general_walk=0, unit_walk=0, update_form=0):
    """Kernel-launch tuning knobs (measurement / tests).  set_tuning() restores the defaults.
    general_walk: reference-order plans on the general walk kernel; unit_walk: the reference-order kernels walk units of
    four rows (C++ loops) instead of group streams (assembly loops); update_form: where forward_update applies the layer
    update -- 0 the library's choice (3 where it fits and the graph has 10+ steps a row, else 1), 1 in the kernel's tail, 3 beside the
    walk with the rows passing through LDS (forward_update returns None where it does not fit; 2 -- rows by reference -- was removed
    in ABI 6 and always answers None)."""
    t = _lib.Tuning(int(threads), int(grid), int(rel_lds), int(x_lds), int(unroll),
                    (ctypes.c_int32 * 3)(int(general_walk), int(unit_walk), int(update_form)))
    check(lib.ultra_set_tuning(ctypes.byref(t)))
