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


def load_triples_dir(root, relation_graph=True):
    """A transductive dataset from raw triple files -- `root` holds train.txt / valid.txt / test.txt with one
    tab- (or space-) separated `head relation tail` triple per line, optionally entities.dict / relations.dict (`id name`
    per line), the layout of kg-datasets/FB15k-237 (PyG RelLinkPredDataset raw files, ultra/datasets.py:186-205).  Returns
    the TEST split in the reference's format: fact graph = training triples + their inverses (edge_type r + R),
    num_relations = 2 R, targets = test triples; `target_triples` as (h, t, r) rows like ultra_amd.synthetic.make_kg.
    `filtered_data` holds the transductive filtering graph of script/run.py:286-288 -- the target triples of ALL three
    splits, no inverses -- which ultra_amd.eval.evaluate uses for the filtered ranking when the caller passes none."""
    import os

    def read_dict(name):
        path = os.path.join(root, name)
        if not os.path.exists(path):
            return None
        vocab = {}
        with open(path) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2:
                    vocab[parts[1]] = int(parts[0])
        return vocab

    ent, rel = read_dict("entities.dict"), read_dict("relations.dict")
    grow_ent, grow_rel = ent is None, rel is None
    ent, rel = ent or {}, rel or {}

    def read_split(name):
        path = os.path.join(root, name)
        if not os.path.exists(path):
            raise FileNotFoundError("%s: no %s (expected train.txt, valid.txt, test.txt)" % (root, name))
        rows = []
        with open(path) as f:
            for line in f:
                parts = line.split()
                if len(parts) != 3:
                    continue
                h, r, t = parts
                for tok, vocab, grow in ((h, ent, grow_ent), (t, ent, grow_ent)):
                    if tok not in vocab:
                        if not grow:
                            raise KeyError("entity %r of %s is not in entities.dict" % (tok, name))
                        vocab[tok] = len(vocab)
                if r not in rel:
                    if not grow_rel:
                        raise KeyError("relation %r of %s is not in relations.dict" % (r, name))
                    rel[r] = len(rel)
                rows.append((ent[h], ent[t], rel[r]))
        return torch.tensor(rows, dtype=torch.long).view(-1, 3)

    train, valid, test = read_split("train.txt"), read_split("valid.txt"), read_split("test.txt")
    R = len(rel)
    edge_index = torch.stack([torch.cat([train[:, 0], train[:, 1]]), torch.cat([train[:, 1], train[:, 0]])])
    edge_type = torch.cat([train[:, 2], train[:, 2] + R])
    everything = torch.cat([train, valid, test])
    data = Data(edge_index=edge_index, edge_type=edge_type, num_nodes=len(ent), num_relations=2 * R,
                target_edge_index=test[:, :2].t().contiguous(), target_edge_type=test[:, 2].contiguous(),
                target_triples=test.contiguous(), valid_triples=valid.contiguous(),
                filtered_data=Data(edge_index=everything[:, :2].t().contiguous(), edge_type=everything[:, 2].contiguous(),
                                   num_nodes=len(ent), num_relations=2 * R))
    if relation_graph:
        from . import tasks
        tasks.build_relation_graph(data)
    return data
