"""Minimal graph container, duck-type compatible with the fields of torch_geometric.data.Data that the
hot path touches (datasets.py:189-197, tasks.py:191-198): edge_index, edge_type, num_nodes, num_edges,
num_relations, relation_graph, target_edge_index, target_edge_type.  PyG itself is not a dependency."""
import copy

import torch


class Data(object):
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_edges(self):
        return self.edge_index.shape[1]

    @property
    def device(self):
        return self.edge_index.device

    def __copy__(self):
        out = self.__class__.__new__(self.__class__)
        out.__dict__.update(self.__dict__)
        return out

    def clone(self):
        return copy.copy(self)

    def to(self, device):
        out = copy.copy(self)
        for k, v in list(out.__dict__.items()):
            if torch.is_tensor(v):
                out.__dict__[k] = v.to(device)
            elif isinstance(v, Data):
                out.__dict__[k] = v.to(device)
        return out

    def keys(self):
        return list(self.__dict__.keys())

    def __repr__(self):
        parts = []
        for k, v in self.__dict__.items():
            parts.append("%s=%s" % (k, list(v.shape) if torch.is_tensor(v) else
                                    ("Data(...)" if isinstance(v, Data) else v)))
        return "Data(%s)" % ", ".join(parts)
