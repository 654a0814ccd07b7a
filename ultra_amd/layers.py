"""GeneralizedRelationalConv -- the relational message-passing layer (reference: ultra/layers.py).

Same constructor, forward signature, parameter names and numerics as the reference layer, without
PyG: the fused path (layers.py:183-231) calls the HIP rspmm engine directly on the module-level
batch-major layout (batch, N, dim), so the three `transpose(0, 1).flatten(1)` copies of the
reference (layers.py:190-192) and the per-call edge sort disappear, and under no_grad the boundary
epilogue (layers.py:199-207) is fused into the kernel.  The unfused message/aggregate path
(layers.py:135-181; `rotate`, or differentiable edge weights) is kept as plain torch index ops.
"""
import os

import torch
from torch import nn
from torch.nn import functional as F

from . import dense, rspmm


# Layer 0 of every NBFNet reads the boundary condition itself: zero everywhere except the head row of each sample.
# With sum aggregation and DistMult messages only that row's out-edges contribute (exactly), so the layer can ask
# the engine for the row-sparse forward instead of the dense one.  Module-level switch for A/B tests.
ONEHOT_FAST_PATH = True
# the boundary condition kept as (row, value) per sample instead of a (batch, N, d) tensor; layer 0 computed on its
# special rows only (A/B switch for tests)
POINT_BOUNDARY_FAST_PATH = True
# ... and under autograd (fine-tuning): the sum aggregate's differentiable rspmm reads the closed form too (A/B switch for tests)
POINT_BOUNDARY_TRAINING = True
# aggregate + update of a training step's layer as one autograd node (A/B switch for tests)
TRAINING_LAYER_NODE = True
# the last layer of a training step evaluated at the rows the readout reads.  Its backward is a pair of gathers in a fixed order
# (ultra_rspmm_rows_backward_gather; round 5 scattered with float atomics: dense.ROWS_BACKWARD_GATHER = False is that route);
# ULTRA_LAST_LAYER_ON_ROWS=0 (or the attribute) runs the whole layer's walk instead
LAST_LAYER_ON_ROWS = os.environ.get("ULTRA_LAST_LAYER_ON_ROWS", "1") != "0"
# aggregate + update of a layer in one launch on dense-format plans (A/B switch for tests)
FUSED_DENSE_LAYER = True
# aggregate + update of a layer in one launch on the reference-order plan of a sparse graph: the update runs in the tail of
# the rspmm kernel, on the rows each workgroup has just summed.  Same bits as the two launches, 1 - 2.7 % faster on the
# benchmark step.  ULTRA_FUSED_SPARSE_LAYER=0 (or the attribute, for A/B tests) selects the two launches.
FUSED_SPARSE_LAYER = os.environ.get("ULTRA_FUSED_SPARSE_LAYER", "1") != "0"


class PointBoundary(object):
    """The NBFNet boundary condition (models.py:59-66, 135-141) in closed form: zero everywhere except row rows[b]
    of sample b, which holds values[b]."""

    def __init__(self, rows, values, num_node):
        # (made contiguous once here: every layer hands these pointers to a kernel)
        self.rows, self.values, self.num_node = rows.to(torch.int64).contiguous(), values.contiguous(), num_node
        self._dense = None

    @property
    def requires_grad(self):
        return self.values.requires_grad

    def dense(self):
        """The (batch, num_node, dim) tensor the reference builds with zeros + scatter_add_ (built once per boundary
        where no gradient flows through it)."""
        if self._dense is not None:
            return self._dense
        if dense.boundary_supported(self.rows, self.values):
            out = dense.onehot_boundary(self.rows, self.values, self.num_node, self.values.shape[-1])
        else:
            out = torch.zeros(len(self.rows), self.num_node, self.values.shape[-1], device=self.values.device,
                              dtype=self.values.dtype)
            index = self.rows.view(-1, 1, 1).expand(-1, 1, self.values.shape[-1])
            out = out.scatter_add_(1, index, self.values.unsqueeze(1))
        if not (torch.is_grad_enabled() and self.values.requires_grad):
            self._dense = out
        return out


def _scatter(src, index, dim_size, reduce):
    # src (batch, M, d) scattered along dim 1 (what torch_scatter.scatter does at layers.py:165-179)
    shape = (src.shape[0], dim_size, src.shape[2])
    idx = index.view(1, -1, 1).expand_as(src)
    if reduce in ("sum", "add"):
        return torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(1, idx, src)
    if reduce == "mean":
        total = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(1, idx, src)
        count = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(1, idx, torch.ones_like(src))
        return total / count.clamp(min=1)
    if reduce in ("max", "min"):
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
        return out.scatter_reduce(1, idx, src, reduce="amax" if reduce == "max" else "amin", include_self=False)
    raise ValueError("Unknown aggregation function `%s`" % reduce)


class GeneralizedRelationalConv(nn.Module):

    eps = 1e-6

    message2mul = {
        "transe": "add",
        "distmult": "mul",
    }

    def __init__(self, input_dim, output_dim, num_relation, query_input_dim, message_func="distmult",
                 aggregate_func="pna", layer_norm=False, activation="relu", dependent=False, project_relations=False):
        super(GeneralizedRelationalConv, self).__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.num_relation = num_relation
        self.query_input_dim = query_input_dim
        self.message_func = message_func
        self.aggregate_func = aggregate_func
        self.dependent = dependent
        self.project_relations = project_relations
        self.node_dim = -2

        if layer_norm:
            self.layer_norm = nn.LayerNorm(output_dim)
        else:
            self.layer_norm = None
        if isinstance(activation, str):
            self.activation = getattr(F, activation)
        else:
            self.activation = activation

        if self.aggregate_func == "pna":
            self.linear = nn.Linear(input_dim * 13, output_dim)
        else:
            self.linear = nn.Linear(input_dim * 2, output_dim)

        if dependent:
            self.relation_linear = nn.Linear(query_input_dim, num_relation * input_dim)
        else:
            if not self.project_relations:
                self.relation = nn.Embedding(num_relation, input_dim)
            else:
                # set by EntityNBFNet.forward from the relation-graph pass (models.py:184-185)
                self.relation = None
                self.relation_projection = nn.Sequential(
                    nn.Linear(input_dim, input_dim),
                    nn.ReLU(),
                    nn.Linear(input_dim, input_dim)
                )

    def forward(self, input, query, boundary, edge_index, edge_type, size, edge_weight=None):
        return self._forward_impl(input, query, boundary, edge_index, edge_type, size, edge_weight, residual=False)

    def _forward_impl(self, input, query, boundary, edge_index, edge_type, size, edge_weight=None, residual=False,
                      relation=None, onehot_rows=None, edge_keep=False):
        """forward() plus the option to fuse the caller's residual `hidden + layer_input` (models.py:158-160),
        to take this layer's relation features precomputed by the caller, to be told that `input` is zero
        outside row onehot_rows[b] of every sample (the layer-0 boundary condition, models.py:139-141), and that
        `edge_weight` is a 0/1 keep mask (edge_keep=True: edges with 0 are absent, base_nbfnet.py:54-77)."""
        batch_size = len(query)

        if relation is None:
            relation = self._relation_for(query, batch_size)
        # edge_weight=None means "all ones" (what every caller on the fused path passes, models.py:143):
        # the kernel then skips the weight stream instead of multiplying by 1.
        return self.propagate(input=input, relation=relation, boundary=boundary, edge_index=edge_index,
                              edge_type=edge_type, size=size, edge_weight=edge_weight, residual=residual,
                              onehot_rows=onehot_rows, edge_keep=edge_keep)

    def _relation_for(self, query, batch_size):
        if self.dependent:
            return self.relation_linear(query).view(batch_size, self.num_relation, self.input_dim)
        if not self.project_relations:
            return self.relation.weight.expand(batch_size, -1, -1)   # stride-0 view, never materialised
        return self.relation_projection(self.relation)

    def layer0_point_supported(self, point, relation, edge_weight):
        """Can ultra_nbf_layer0 run this layer on a one-hot input?  (sum / DistMult, 64-d, fp32, inference)"""
        return (POINT_BOUNDARY_FAST_PATH and point.values.is_cuda and point.values.dtype == torch.float32
                and not torch.is_grad_enabled() and self.aggregate_func in ("sum", "max") and self.message_func == "distmult"
                and self.input_dim == 64 and self.output_dim == 64 and self.linear.in_features == 128
                and (self.activation is None or self.activation is F.relu)
                and (relation is None or (relation.dtype == torch.float32 and relation.shape[-1] == 64))
                and (edge_weight is None or not edge_weight.requires_grad))

    def layer0_fill(self, edge_index, edge_type, num_node, num_relation, batch_size):
        """The constant rows of forward_layer0_point (every row that layer 0 cannot reach: relu(LayerNorm(bias))) as a fresh
        (batch, N, 64) tensor on the CURRENT stream -- they depend on this layer's parameters only, so a model may launch
        them beside whatever produces the layer's other operands and pass the tensor as `out`."""
        plan = rspmm.get_plan(edge_index, edge_type, num_node, num_relation)
        return plan.layer0_fill(batch_size, self.linear, self.layer_norm, relu=self.activation is not None,
                                device=edge_index.device)

    def forward_layer0_point(self, point, query, edge_index, edge_type, num_node, edge_weight=None, residual=False,
                             relation=None, out=None):
        """This layer applied to the boundary condition itself (what layer 0 of every NBFNet does, models.py:72-80,
        150-163), evaluated only where the result differs from relu(LayerNorm(bias))."""
        batch_size = len(point.rows)
        if relation is None:
            relation = self._relation_for(query, batch_size)
        plan = rspmm.get_plan(edge_index, edge_type, num_node, relation.shape[1])
        return plan.layer0(relation, point.rows, point.values, self.linear, self.layer_norm,
                           relu=self.activation is not None, residual=residual, edge_weight=edge_weight,
                           aggregate=self.aggregate_func, out=out)

    def propagate(self, edge_index, size=None, residual=False, onehot_rows=None, edge_keep=False, **kwargs):
        edge_weight = kwargs["edge_weight"]
        if edge_keep and self.message_func == "rotate":
            # the unfused scatter path needs the edges really gone: the caller removes them
            raise RuntimeError("edge_keep masks serve the fused TransE / DistMult path")
        if isinstance(kwargs["boundary"], PointBoundary) and (
                (edge_weight is not None and edge_weight.requires_grad) or self.message_func == "rotate"
                or self.aggregate_func not in ("sum", "max") or not kwargs["input"].is_cuda
                or (self.aggregate_func == "max" and (torch.is_grad_enabled() or edge_keep))
                or (torch.is_grad_enabled() and not self.point_boundary_trains()
                    and (kwargs["input"].requires_grad or kwargs["relation"].requires_grad
                         or kwargs["boundary"].requires_grad))):
            kwargs["boundary"] = kwargs["boundary"].dense()     # paths that need the boundary as a tensor
        if (edge_weight is not None and edge_weight.requires_grad) or self.message_func == "rotate":
            # layers.py:91-94: the fused kernel covers TransE / DistMult with constant edge weights only
            out = self._propagate_unfused(edge_index, size, **kwargs)
            return out + kwargs["input"] if residual else out
        num_node = size[0] if size is not None else kwargs["input"].shape[1]
        fused = self._fused_dense_layer(edge_index, kwargs, num_node, residual, onehot_rows)
        if fused is not None:
            return fused
        fused = self._fused_sparse_layer(edge_index, kwargs, num_node, residual, onehot_rows, edge_keep)
        if fused is not None:
            return fused
        fused = self._training_layer(edge_index, kwargs, num_node, residual, onehot_rows, edge_keep)
        if fused is not None:
            return fused
        out = self.message_and_aggregate(edge_index, kwargs["input"], kwargs["relation"], kwargs["boundary"],
                                         kwargs["edge_type"], edge_weight, edge_index[1], num_node,
                                         onehot_rows=onehot_rows, edge_keep=edge_keep)
        return self.update(out, kwargs["input"], residual=residual)

    def point_boundary_trains(self):
        """The differentiable rspmm takes the boundary condition in closed form for the sum aggregate (rspmm._PlanRSPMM,
        rspmm._OnehotRSPMM): no (batch, N, d) boundary gradient is formed, layer 0 walks the sources' edges only."""
        return POINT_BOUNDARY_TRAINING and self.aggregate_func == "sum" and self.message_func in self.message2mul

    def _fused_dense_layer(self, edge_index, kwargs, num_node, residual, onehot_rows):
        """Aggregate + update in one launch where the graph has a dense-format plan (ULTRA's relation graph)."""
        input, relation, boundary = kwargs["input"], kwargs["relation"], kwargs["boundary"]
        if not (FUSED_DENSE_LAYER and kwargs["edge_weight"] is None and self.aggregate_func == "sum"
                and self.message_func == "distmult" and input.is_cuda and not torch.is_grad_enabled()
                and onehot_rows is None and dense.conv_update_supported(self, input, input)):
            return None
        plan = rspmm.get_plan(edge_index, kwargs["edge_type"], num_node, relation.shape[1])
        if plan.dense is None:
            return None
        point = (boundary.rows, boundary.values) if isinstance(boundary, PointBoundary) else None
        return plan.fused_layer(relation, input, self.linear, self.layer_norm, relu=self.activation is not None,
                                residual=residual, boundary=None if point is not None else boundary, point=point)

    def _order_free(self, edge_index, num_node):
        """max (and min) do not depend on the order of their operands (but for the sign of a zero), so a graph whose rows are
        mostly long -- ULTRA's relation graph: every row has hundreds of edges, each a serial chain on the reference-order
        kernels -- goes to the re-associating plan, which splits such rows over many lanes, and still returns the
        reference's values."""
        return self.aggregate_func == "max" and edge_index.shape[1] >= 128 * max(int(num_node), 1)

    def _fused_sparse_layer(self, edge_index, kwargs, num_node, residual, onehot_rows, edge_keep):
        """Aggregate + update in one launch on the reference-order plan of a sparse graph (the entity graph): the workgroup
        that sums a row also applies the layer update to it (ultra_rspmm_forward_update).  Same bits as the two launches."""
        input, relation, boundary = kwargs["input"], kwargs["relation"], kwargs["boundary"]
        if not (FUSED_SPARSE_LAYER and kwargs["edge_weight"] is None and self.aggregate_func in ("sum", "max")
                and self.message_func in self.message2mul and input.is_cuda and not torch.is_grad_enabled()
                and onehot_rows is None and not edge_keep and isinstance(boundary, PointBoundary)
                and input.dim() == 3 and relation.dtype == torch.float32 and dense.conv_update_supported(self, input, input)
                and not self._order_free(edge_index, num_node)):
            return None
        plan = rspmm.get_plan(edge_index, kwargs["edge_type"], num_node, relation.shape[1])
        ln = self.layer_norm
        flags = (dense.CONV_LAYER_NORM if ln is not None else 0) | (dense.CONV_RELU if self.activation is not None else 0) \
            | (dense.CONV_RESIDUAL if residual else 0)
        return plan.forward_update(relation, input, self.linear.weight, self.linear.bias, ln.weight if ln is not None else None,
                                   ln.bias if ln is not None else None, float(ln.eps) if ln is not None else 1e-5, flags,
                                   mul=self.message2mul[self.message_func], point=(boundary.rows, boundary.values),
                                   sum="add" if self.aggregate_func == "sum" else "max")

    def _training_layer(self, edge_index, kwargs, num_node, residual, onehot_rows, edge_keep):
        """Aggregate + update of a training step's layer as one autograd node (dense.TrainLayerFunction): the layer input's two
        gradient shares -- through the rspmm and through the update -- leave the backward already summed.  Sum aggregate,
        TransE / DistMult, the ULTRA update shape; layer 0 (one-hot input) has its own route in message_and_aggregate."""
        input, relation, boundary, edge_weight = kwargs["input"], kwargs["relation"], kwargs["boundary"], kwargs["edge_weight"]
        point = boundary if isinstance(boundary, PointBoundary) else None
        if not (TRAINING_LAYER_NODE and torch.is_grad_enabled() and onehot_rows is None and self.aggregate_func == "sum"
                and self.message_func in self.message2mul and input.is_cuda and input.dim() == 3
                and input.dtype == torch.float32 and relation.dtype == torch.float32
                and (edge_weight is None or (not edge_weight.requires_grad and edge_weight.dtype == torch.float32))
                and (input.requires_grad or relation.requires_grad or boundary.requires_grad)
                and (point is not None or boundary.dtype == torch.float32)
                and dense.conv_update_supported(self, input, input)):
            return None
        plan = rspmm.get_plan(edge_index, kwargs["edge_type"], num_node, relation.shape[1], exact_order=False)
        ln = self.layer_norm
        flags = (dense.CONV_LAYER_NORM if ln is not None else 0) | (dense.CONV_RELU if self.activation is not None else 0) \
            | (dense.CONV_RESIDUAL if residual else 0)
        return dense.TrainLayerFunction.apply(
            plan, self.message2mul[self.message_func], bool(edge_keep), float(ln.eps) if ln is not None else 1e-5, flags,
            edge_weight, relation, input, None if point is not None else boundary,
            point.rows if point is not None else None, point.values if point is not None else None,
            self.linear.weight, self.linear.bias, ln.weight if ln is not None else None, ln.bias if ln is not None else None)

    def training_rows_layer(self, input, query, boundary, edge_index, edge_type, num_node, rows, edge_weight=None, residual=False,
                            relation=None):
        """This layer's output at the listed rows only -- (batch, n_list, 64) for rows (batch, n_list) -- as one autograd node
        (dense.TrainRowsLayerFunction): what the LAST layer of a training step needs, since the readout reads the candidates'
        rows alone (models.py:202-207).  None where the route does not apply (the caller then runs the whole layer)."""
        if relation is None:
            relation = self._relation_for(query, len(query))
        point = boundary if isinstance(boundary, PointBoundary) else None
        if not (LAST_LAYER_ON_ROWS and torch.is_grad_enabled() and self.aggregate_func == "sum"
                and self.message_func in self.message2mul and input.is_cuda and input.dim() == 3 and rows.dim() == 2
                and 4 * rows.shape[1] <= num_node         # (a list that covers most of the graph: the whole layer's walk is the better one)
                and input.dtype == torch.float32 and relation.dtype == torch.float32
                and (edge_weight is None or (not edge_weight.requires_grad and edge_weight.dtype == torch.float32))
                and (input.requires_grad or relation.requires_grad or boundary.requires_grad)
                and (point is not None or (boundary.dtype == torch.float32 and not boundary.requires_grad))
                and (point is None or point.values.dtype == torch.float32)
                and dense.conv_update_supported(self, input, input)):
            return None
        plan = rspmm.get_plan(edge_index, edge_type, num_node, relation.shape[1], exact_order=False)
        ln = self.layer_norm
        flags = (dense.CONV_LAYER_NORM if ln is not None else 0) | (dense.CONV_RELU if self.activation is not None else 0) \
            | (dense.CONV_RESIDUAL if residual else 0)
        return dense.TrainRowsLayerFunction.apply(
            plan, self.message2mul[self.message_func], float(ln.eps) if ln is not None else 1e-5, flags, edge_weight, relation, input,
            rows, None if point is not None else boundary, point.rows if point is not None else None,
            point.values if point is not None else None, self.linear.weight, self.linear.bias,
            ln.weight if ln is not None else None, ln.bias if ln is not None else None)

    # ---- unfused path: gather edge_index[0], scatter to edge_index[1] -- PyG's direction (layers.py:135-181) ----
    def _propagate_unfused(self, edge_index, size, input, relation, boundary, edge_type, edge_weight):
        num_target = size[1] if size is not None else input.shape[1]
        if edge_weight is None:
            edge_weight = torch.ones(edge_index.shape[1], device=input.device, dtype=input.dtype)
        message = self.message(input.index_select(self.node_dim, edge_index[0]), relation, boundary, edge_type)
        return self.update(self.aggregate(message, edge_weight, edge_index[1], num_target), input)

    @staticmethod
    def _combine(message_func, source, relation):
        """One message per edge from its source state and relation feature (layers.py:138-151)."""
        if message_func == "transe":
            return source + relation
        if message_func == "distmult":
            return source * relation
        if message_func == "rotate":         # complex product on (real | imaginary) halves
            s_re, s_im = source.chunk(2, dim=-1)
            r_re, r_im = relation.chunk(2, dim=-1)
            return torch.cat([s_re * r_re - s_im * r_im, s_re * r_im + s_im * r_re], dim=-1)
        raise ValueError("Unknown message function `%s`" % message_func)

    def message(self, input_j, relation, boundary, edge_type):
        edges = self._combine(self.message_func, input_j, relation.index_select(self.node_dim, edge_type))
        # the boundary condition rides along as one self-loop message per node (layers.py:153-155)
        return torch.cat([edges, boundary], dim=self.node_dim)

    def aggregate(self, input, edge_weight, index, dim_size):
        # the self-loop messages appended by message(): target = the node itself, weight 1
        loops = torch.arange(dim_size, device=input.device)
        index = torch.cat([index, loops])
        weight = torch.cat([edge_weight, torch.ones(dim_size, device=input.device, dtype=edge_weight.dtype)]).view(1, -1, 1)
        weighted = input * weight
        if self.aggregate_func != "pna":
            return _scatter(weighted, index, dim_size, self.aggregate_func)
        degree = torch.bincount(index, minlength=dim_size).to(input.dtype).view(1, -1, 1)
        return self._pna(_scatter(weighted, index, dim_size, "mean"), _scatter(input ** 2 * weight, index, dim_size, "mean"),
                         _scatter(weighted, index, dim_size, "max"), _scatter(weighted, index, dim_size, "min"), degree)

    def _pna(self, mean, sq_mean, maximum, minimum, degree):
        """Principal neighbourhood aggregation (layers.py:165-179, 208-226): [mean, max, min, std] x [1, s, 1 / s] with
        s = log(degree) / mean(log(degree)); (batch, node, 12 d) in the reference's feature order."""
        std = (sq_mean - mean ** 2).clamp(min=self.eps).sqrt()
        stats = torch.stack([mean, maximum, minimum, std], dim=-1).flatten(-2)      # (batch, node, 4 d), statistic fastest
        scale = degree.log()
        scale = scale / scale.mean()
        scalers = torch.cat([torch.ones_like(scale), scale, 1 / scale.clamp(min=1e-2)], dim=-1)      # (1, node, 3)
        return (stats.unsqueeze(-1) * scalers.unsqueeze(-2)).flatten(-2)

    # ---- fused path ----
    def message_and_aggregate(self, edge_index, input, relation, boundary, edge_type, edge_weight, index, dim_size,
                              onehot_rows=None, edge_keep=False):
        """(batch, N, d) in, (batch, N, d') out.  Aggregates into edge_index[0] from edge_index[1]
        (the fused kernel's direction, rspmm.cpp:143-145) -- not the unfused path's direction."""
        batch_size, num_node = input.shape[:2]
        if self.message_func in self.message2mul:
            mul = self.message2mul[self.message_func]
        else:
            raise ValueError("Unknown message function `%s`" % self.message_func)
        if edge_weight is not None and not torch.is_floating_point(edge_weight):
            edge_weight = edge_weight.to(input.dtype)
        needs_grad = torch.is_grad_enabled() and (input.requires_grad or relation.requires_grad or
                                                   boundary.requires_grad)
        # a differentiable call (training step) takes the re-associating plan: its backward is a scatter-add and its
        # result feeds a stochastic optimiser step -- there is no reference summation order to reproduce
        plan = rspmm.get_plan(edge_index, edge_type, num_node, relation.shape[1],
                              exact_order=False if (needs_grad or self._order_free(edge_index, num_node)) else None)

        point, point_boundary = None, None
        if isinstance(boundary, PointBoundary):     # (propagate() only lets it through for the fused sum / max paths)
            point, point_boundary, boundary = (boundary.rows, boundary.values), boundary, None

        def agg(sum, rel=relation, x=input, fuse_boundary=None):
            if point is not None and needs_grad:
                # (propagate() lets the closed form through under autograd for the sum aggregate only)
                return rspmm.plan_rspmm(plan, rel, x, edge_weight, sum=sum, mul=mul, keep=edge_keep, point=point)
            if point is not None:
                out = plan.forward(rel, x, edge_weight=edge_weight, sum=sum, mul=mul, point=point)
                if out is not None:
                    return out
                # (a plan that does not serve the point form under min / max: the boundary as a tensor)
                return plan.forward(rel, x, edge_weight=edge_weight, boundary=point_boundary.dense(), sum=sum, mul=mul)
            if needs_grad:
                if sum == "add":      # boundary added in the kernel's epilogue; its gradient is the output gradient
                    return rspmm.plan_rspmm(plan, rel, x, edge_weight, sum=sum, mul=mul, boundary=fuse_boundary,
                                            keep=edge_keep)
                out = rspmm.plan_rspmm(plan, rel, x, edge_weight, sum=sum, mul=mul, keep=edge_keep)
                if fuse_boundary is None:
                    return out
                return torch.max(out, fuse_boundary) if sum == "max" else torch.min(out, fuse_boundary)
            return plan.forward(rel, x, edge_weight=edge_weight, boundary=fuse_boundary, sum=sum, mul=mul, keep=edge_keep)

        if self.aggregate_func in ("mean", "pna"):
            # layers.py:193 -- PyG's `index` is edge_index[1]; under a keep mask the degree counts the kept edges only
            # (what bincount gives on the reference's filtered copy of the graph, base_nbfnet.py:54-77)
            if edge_keep and edge_weight is not None:
                degree_out = torch.zeros(dim_size, dtype=input.dtype, device=input.device).index_add_(
                    0, index, edge_weight.detach().to(input.dtype))
            else:
                degree_out = torch.bincount(index, minlength=dim_size).to(input.dtype)
            degree_out = (degree_out + 1).view(1, -1, 1)

        if (ONEHOT_FAST_PATH and onehot_rows is not None and needs_grad and point is not None and self.aggregate_func == "sum"
                and mul == "mul" and input.is_cuda and input.dtype == torch.float32 and dim_size == input.shape[1]
                and (edge_weight is None or not edge_weight.requires_grad)):
            # layer 0 of a training step: `input` IS the boundary condition as a tensor (the caller's contract for
            # onehot_rows beside a PointBoundary) -- forward and backward visit the sources' out-edges only
            update = rspmm.onehot_rspmm(plan, edge_index, edge_type, relation, point[0], point[1], input, edge_weight)
        elif (ONEHOT_FAST_PATH and onehot_rows is not None and not needs_grad and self.aggregate_func == "sum"
                and mul == "mul" and input.is_cuda and point is None):
            # row-sparse input (layer 0): only the edges leaving the source rows contribute to a sum of products
            update = plan.forward_onehot(relation, input, onehot_rows, edge_weight=edge_weight, boundary=boundary)
        elif self.aggregate_func == "sum":
            update = agg("add", fuse_boundary=boundary)
        elif self.aggregate_func == "mean":
            update = agg("add", fuse_boundary=boundary) / degree_out
        elif self.aggregate_func == "max":
            update = agg("max", fuse_boundary=boundary)
        elif self.aggregate_func == "pna":
            # four rspmm calls: sum and sum of squares (relation and input squared: (r x)^2 = r^2 x^2), max, min
            total = agg("add")
            sq_total = agg("add", rel=relation ** 2, x=input ** 2)
            update = self._pna((total + boundary) / degree_out, (sq_total + boundary ** 2) / degree_out,
                               agg("max", fuse_boundary=boundary), agg("min", fuse_boundary=boundary), degree_out)
        else:
            raise ValueError("Unknown aggregation function `%s`" % self.aggregate_func)
        return update

    def update(self, update, input, residual=False):
        if dense.conv_update_supported(self, input, update):
            # one MFMA kernel: Linear(cat[input, update]) -> LayerNorm -> ReLU (-> + input)
            return dense.conv_update(self, input, update, residual)
        output = self.linear(torch.cat([input, update], dim=-1))
        if self.layer_norm:
            output = self.layer_norm(output)
        if self.activation:
            output = self.activation(output)
        if residual:
            output = output + input
        return output
