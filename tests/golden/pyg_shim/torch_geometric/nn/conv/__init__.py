"""MessagePassing stub: exactly what ultra/layers.py:90-133 touches, flow = source_to_target."""
import inspect
from collections import OrderedDict

import torch


class _Inspector(object):
    def __init__(self, module):
        self.module = module

    def collect_param_data(self, name, data):
        params = list(inspect.signature(getattr(self.module, name)).parameters)
        if name in ("aggregate", "update", "message_and_aggregate"):
            params = params[1:]   # PyG inspects these with exclude=[0]: the first argument is passed positionally
        return {k: data[k] for k in params if k in data}

    distribute = collect_param_data


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        self.node_dim = node_dim
        self.inspector = _Inspector(self)
        self._fused_user_args = None
        self._propagate_forward_pre_hooks = OrderedDict()
        self._propagate_forward_hooks = OrderedDict()
        self._message_and_aggregate_forward_pre_hooks = OrderedDict()
        self._message_and_aggregate_forward_hooks = OrderedDict()

    def _check_input(self, edge_index, size):
        return list(size) if size is not None else [None, None]

    def _collect(self, args, edge_index, size, kwargs):
        out = dict(kwargs)
        for key, value in kwargs.items():
            if torch.is_tensor(value) and value.dim() >= 2 and value.size(self.node_dim) in (size[0], size[1]) \
                    and key in ("input",):
                out[key + "_j"] = value.index_select(self.node_dim, edge_index[0])
                out[key + "_i"] = value.index_select(self.node_dim, edge_index[1])
        out["edge_index"] = edge_index
        out["index"] = edge_index[1]
        out["dim_size"] = size[1]
        out["size"] = size
        return out

    def propagate(self, edge_index, size=None, **kwargs):
        size = self._check_input(edge_index, size)
        coll = self._collect(None, edge_index, size, kwargs)
        msg = self.message(**self.inspector.collect_param_data("message", coll))
        out = self.aggregate(msg, **self.inspector.collect_param_data("aggregate", coll))
        return self.update(out, **self.inspector.collect_param_data("update", coll))
