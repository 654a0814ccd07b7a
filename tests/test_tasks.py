"""Evaluation glue (ultra_amd/tasks.py) against golden outputs of the reference's ultra/tasks.py. CPU only."""
import os

import pytest
import torch

from tests.test_oracle_model import MODELS, load_golden
from ultra_amd import synthetic, tasks
from ultra_amd.data import Data


@pytest.mark.parametrize("ckpt,aggr", MODELS[:1])
def test_tasks_match_reference(ckpt, aggr):
    g, _, data, _ = load_golden(ckpt, aggr)
    batch = g["batch"]
    t_batch, h_batch = tasks.all_negative(data, batch)
    assert torch.equal(t_batch, g["t_batch"]) and torch.equal(h_batch, g["h_batch"])
    t_mask, h_mask = tasks.strict_negative_mask(data, batch)
    assert torch.equal(t_mask, g["t_mask"]) and torch.equal(h_mask, g["h_mask"])
    pos_h, pos_t, _ = batch.t()
    assert torch.equal(tasks.compute_ranking(g["t_pred"], pos_t, t_mask), g["t_rank"])
    assert torch.equal(tasks.compute_ranking(g["h_pred"], pos_h, h_mask), g["h_rank"])
    assert torch.equal(tasks.compute_ranking(g["t_pred"], pos_t), ((g["t_pred"].gather(1, pos_t[:, None]) <= g["t_pred"]).sum(-1) + 1))


@pytest.mark.parametrize("ckpt,aggr", MODELS[:1])
def test_build_relation_graph_matches_reference(ckpt, aggr):
    g, _, data, _ = load_golden(ckpt, aggr)
    fresh = Data(edge_index=data.edge_index, edge_type=data.edge_type, num_nodes=data.num_nodes,
                 num_relations=data.num_relations)
    tasks.build_relation_graph(fresh, node_chunk=64)      # several node chunks
    assert torch.equal(fresh.relation_graph.edge_index, g["rel_edge_index"])
    assert torch.equal(fresh.relation_graph.edge_type, g["rel_edge_type"])
    assert fresh.relation_graph.num_nodes == data.num_relations and fresh.relation_graph.num_relations == 4


def test_edge_match_counts_duplicates():
    edge_index = torch.tensor([[0, 0, 1, 0, 2], [1, 1, 2, 3, 0]])
    query = torch.tensor([[0, 2, 1, 4], [1, 0, 2, 4]])
    ids, num = tasks.edge_match(edge_index, query)
    assert num.tolist() == [2, 1, 1, 0]
    assert sorted(ids[:2].tolist()) == [0, 1] and ids[2:].tolist() == [4, 2]


def test_negative_sampling_is_strict():
    data = synthetic.make_kg(num_node=60, num_triple=400, num_relation_base=3, seed=5, relation_graph=False)
    batch = torch.stack([data.edge_index[0, :8], data.edge_index[1, :8], data.edge_type[:8]], dim=-1)
    torch.manual_seed(0)
    out = tasks.negative_sampling(data, batch, 16, strict=True)
    assert out.shape == (8, 17, 3)
    assert torch.equal(out[:, 0], batch)
    t_mask, h_mask = tasks.strict_negative_mask(data, batch)
    for i in range(4):      # first half: corrupted tails are never known true tails
        assert t_mask[i, out[i, 1:, 1]].all() and (out[i, :, 0] == batch[i, 0]).all()
    for i in range(4, 8):   # second half: corrupted heads
        assert h_mask[i, out[i, 1:, 0]].all() and (out[i, :, 1] == batch[i, 1]).all()


def test_prefetched_negatives_are_the_plain_loops_draws():
    """prefetch_negatives (the sampler one batch ahead of the step): the same batches, in the same order, as calling
    negative_sampling in the loop -- on the CPU it IS that loop; an empty loader yields nothing."""
    data = synthetic.make_kg(num_node=60, num_triple=400, num_relation_base=3, seed=5, relation_graph=False)
    triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)
    batches = [triples[8 * i:8 * i + 8] for i in range(5)]
    torch.manual_seed(3)
    plain = [tasks.negative_sampling(data, b, 16, strict=True) for b in batches]
    torch.manual_seed(3)
    ahead = list(tasks.prefetch_negatives(iter(batches), data, 16, strict=True))
    assert len(ahead) == 5 and all(torch.equal(a, b) for a, b in zip(ahead, plain))
    assert list(tasks.prefetch_negatives(iter([]), data, 16)) == []


def test_synthetic_shapes():
    s = synthetic.SHAPES["fb15k237"]
    assert (s["num_node"], 2 * s["num_triple"], 2 * s["num_relation_base"], s["num_test"]) == (14541, 544230, 474, 20466)
    d = synthetic.make_kg(num_node=100, num_triple=500, num_relation_base=4, seed=1)
    assert d.edge_index.shape == (2, 1000) and d.num_relations == 8
    assert torch.equal(d.edge_index[0, :500], d.edge_index[1, 500:])          # inverse edges appended
    assert torch.equal(d.edge_type[500:], d.edge_type[:500] + 4)
    d2 = synthetic.make_kg(num_node=100, num_triple=500, num_relation_base=4, seed=1)
    assert torch.equal(d.edge_index, d2.edge_index)                            # seeded


def _negative_sampling_golden():
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "negative_sampling.pt"))
    data = Data(edge_index=g["edge_index"], edge_type=g["edge_type"], num_nodes=g["num_nodes"], num_relations=g["num_relations"])
    return data, g["cases"]


def test_negative_sampling_replays_the_reference_draws():
    """tests/golden/negative_sampling.pt holds, for five (batch, num_negative, strict) cases on a KG with a hub node, the global
    generator's state right before the REFERENCE's tasks.negative_sampling (tasks.py:42-76) and the batch it returned
    (tests/golden/gen_golden.py: gen_negative_sampling).  From the same state ultra_amd.tasks.negative_sampling returns the same
    batch, id for id (CPU: the masks + nonzero() formulation)."""
    data, cases = _negative_sampling_golden()
    assert len(cases) == 5 and any(not c["strict"] for c in cases)
    for case in cases:
        torch.set_rng_state(case["rng_state"])
        got = tasks.negative_sampling(data, case["batch"], case["num_negative"], strict=case["strict"])
        assert torch.equal(got, case["out"]), (case["num_negative"], case["strict"])
        if case["strict"]:      # (the hub rows are really filtered: no sampled tail is a known answer)
            t_mask, h_mask = tasks.strict_negative_mask(data, case["batch"])
            half = len(case["batch"]) // 2
            assert all(t_mask[i, got[i, 1:, 1]].all() for i in range(half))
            assert all(h_mask[i, got[i, 1:, 0]].all() for i in range(half, len(case["batch"])))


def easy_edges_golden():
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "easy_edges.pt"))
    data = Data(edge_index=g["edge_index"], edge_type=g["edge_type"], num_nodes=g["num_nodes"], num_relations=g["num_relations"])
    return data, g["cases"]


def test_easy_edge_mask_keeps_the_edges_the_reference_keeps():
    """tests/golden/easy_edges.pt: five batches (positives that are graph edges + strict negatives, both halves; a duplicated
    edge; the same pair under another relation) and, per batch, which edges the REFERENCE's BaseNBFNet.remove_easy_edges
    (base_nbfnet.py:54-77) left in the graph (gen_golden.py: gen_easy_edges).  models.BaseNBFNet.easy_edge_mask marks the same."""
    from ultra_amd import models
    data, cases = easy_edges_golden()
    assert {c["remove_one_hop"] for c in cases} == {False, True}
    for case in cases:
        model = models.EntityNBFNet(64, [64] * 2, remove_one_hop=case["remove_one_hop"])
        h, t, r = case["batch"].unbind(-1)
        assert torch.equal(model.easy_edge_mask(data, h, t, r), case["keep"])
        assert 0 < int((~case["keep"]).sum()) < 100
        kept = model.remove_easy_edges(data, h, t, r)
        assert torch.equal(kept.edge_index, data.edge_index[:, case["keep"]])
        assert torch.equal(kept.edge_type, data.edge_type[case["keep"]])
