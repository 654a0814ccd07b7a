"""Seeded synthetic knowledge graphs with the public shapes of the benchmark datasets.

No dataset is on disk and there is no network, so FB15k237 / WN18RR / CoDEx-L / YAGO3-10 are replaced
by random graphs of the same size: power-law heads (p ~ rank^-0.8), uniform tails,
Zipfian relations, inverse edges appended exactly like the reference's datasets do
(edge_index = [[h; t], [t; h]], edge_type = [r; r + R/2], datasets.py:186-197).
"""
import torch

from . import tasks
from .data import Data

# num_node, training triples, base relations (R/2), test triples
SHAPES = {
    "fb15k237": dict(num_node=14541, num_triple=272115, num_relation_base=237, num_test=20466),
    "wn18rr": dict(num_node=40943, num_triple=86835, num_relation_base=11, num_test=3134),
    "codex_l": dict(num_node=77951, num_triple=551193, num_relation_base=69, num_test=30622),
    "yago310": dict(num_node=123182, num_triple=1079040, num_relation_base=37, num_test=5000),
}


def _draw(g, num_node, num_relation_base, count):
    ph = 1.0 / torch.arange(1, num_node + 1, dtype=torch.float64) ** 0.8
    pr = 1.0 / torch.arange(1, num_relation_base + 1, dtype=torch.float64)
    h = torch.multinomial(ph, count, replacement=True, generator=g)
    t = torch.randint(0, num_node, (count,), generator=g)
    r = torch.multinomial(pr, count, replacement=True, generator=g)
    return h, t, r


def make_kg(num_node, num_triple, num_relation_base, num_test=None, seed=1234, relation_graph=True):
    g = torch.Generator().manual_seed(seed)
    h, t, r = _draw(g, num_node, num_relation_base, num_triple)
    edge_index = torch.stack([torch.cat([h, t]), torch.cat([t, h])])
    edge_type = torch.cat([r, r + num_relation_base])
    if num_test is None:
        num_test = max(8, min(num_triple // 10, 4096))
    th, tt, tr = _draw(g, num_node, num_relation_base, num_test)
    data = Data(edge_index=edge_index, edge_type=edge_type, num_nodes=num_node,
                num_relations=2 * num_relation_base,
                target_edge_index=torch.stack([th, tt]), target_edge_type=tr,
                target_triples=torch.stack([th, tt, tr], dim=-1))
    if relation_graph:
        tasks.build_relation_graph(data)
    return data


def to_device(data, device):
    return data.to(device)


def default_model_cfg(aggregate_func="sum", message_func="distmult"):
    """config/transductive/inference.yaml:9-24 -- both NBFNets 6 x 64, distmult, sum, short_cut, layer_norm."""
    def one(cls):
        return {"class": cls, "input_dim": 64, "hidden_dims": [64] * 6, "message_func": message_func,
                "aggregate_func": aggregate_func, "short_cut": True, "layer_norm": True}
    return {"rel_model_cfg": one("RelNBFNet"), "entity_model_cfg": one("EntityNBFNet")}
