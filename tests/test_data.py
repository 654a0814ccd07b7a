"""Raw triple files in the kg-datasets layout (ultra/datasets.py:186-205: train.txt / valid.txt / test.txt, one
`head relation tail` per line) against the in-memory graph they were written from (tests/golden/gen_kg_fixture.py): fact
graph = training triples + inverses with relation ids r and r + R, targets = the test split, and the transductive
filtering graph of script/run.py:286-288 (every split's triples, no inverses)."""
import os
import shutil

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "kg_fixture")


def _memory_graph():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_kg_fixture", os.path.join(ROOT, "tests", "golden", "gen_kg_fixture.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    from ultra_amd import synthetic
    data = synthetic.make_kg(num_node=gen.NUM_NODE, num_triple=gen.NUM_TRAIN, num_relation_base=gen.NUM_REL,
                             num_test=gen.NUM_VALID + gen.NUM_TEST, seed=gen.SEED, relation_graph=True)
    return gen, data


def test_reader_reproduces_the_graph_the_files_were_written_from():
    from ultra_amd.data import load_triples_dir
    gen, mem = _memory_graph()
    data = load_triples_dir(FIXTURE)
    assert data.num_nodes == gen.NUM_NODE and data.num_relations == 2 * gen.NUM_REL
    # fact graph: training triples, then their inverses with relation ids r + R (datasets.py:189-197 via PyG's RelLinkPredDataset)
    assert torch.equal(data.edge_index, mem.edge_index) and torch.equal(data.edge_type, mem.edge_type)
    assert int(data.edge_type[:gen.NUM_TRAIN].max()) < gen.NUM_REL <= int(data.edge_type[gen.NUM_TRAIN:].min())
    held = mem.target_triples
    assert torch.equal(data.valid_triples, held[:gen.NUM_VALID])
    assert torch.equal(data.target_triples, held[gen.NUM_VALID:])
    assert torch.equal(data.target_edge_index, held[gen.NUM_VALID:, :2].t()) and torch.equal(data.target_edge_type, held[gen.NUM_VALID:, 2])
    # relation graph of the fact graph (tasks.py:144-199)
    assert torch.equal(data.relation_graph.edge_index, mem.relation_graph.edge_index)
    assert torch.equal(data.relation_graph.edge_type, mem.relation_graph.edge_type)
    # filtering graph: all three splits' (h, t, r), no inverses (script/run.py:286-288)
    f = data.filtered_data
    train = torch.stack([mem.edge_index[0, :gen.NUM_TRAIN], mem.edge_index[1, :gen.NUM_TRAIN], mem.edge_type[:gen.NUM_TRAIN]], dim=-1)
    want = torch.cat([train, held])
    assert torch.equal(torch.cat([f.edge_index.t(), f.edge_type.unsqueeze(1)], dim=1), want)
    assert f.num_nodes == gen.NUM_NODE


def test_reader_without_dictionaries_numbers_names_in_order_of_appearance(tmp_path):
    from ultra_amd.data import load_triples_dir
    for name in ("train.txt", "valid.txt", "test.txt"):
        shutil.copy(os.path.join(FIXTURE, name), tmp_path / name)
    with_dict, plain = load_triples_dir(FIXTURE, relation_graph=False), load_triples_dir(str(tmp_path), relation_graph=False)
    assert plain.num_edges == with_dict.num_edges and plain.num_relations == with_dict.num_relations
    assert plain.num_nodes <= with_dict.num_nodes          # (entities that occur in no triple have no id here)
    # the same graph up to the renumbering: ids in order of first appearance, inverse edges mirrored
    first = [l.split() for l in open(tmp_path / "train.txt").read().splitlines()[:3]]
    assert plain.edge_index[0, 0] == 0 and plain.edge_type[0] == 0
    E = plain.num_edges // 2
    assert torch.equal(plain.edge_index[:, :E], plain.edge_index[:, E:].flip(0))
    assert torch.equal(plain.edge_type[:E] + plain.num_relations // 2, plain.edge_type[E:])
    assert sorted(torch.bincount(plain.edge_type).tolist()) == sorted(torch.bincount(with_dict.edge_type).tolist())
    assert len(first) == 3
    with pytest.raises(FileNotFoundError):
        load_triples_dir(str(tmp_path / "missing"))


def test_evaluation_filters_against_every_split_of_a_dataset_read_from_files():
    """evaluate(model, data) on a dataset from load_triples_dir ranks against the train + valid + test filter
    (script/run.py:286-288), not against the fact graph alone."""
    from tests.test_distributed import StubScorer
    from ultra_amd import eval as ueval
    from ultra_amd.data import Data, load_triples_dir
    data = load_triples_dir(FIXTURE, relation_graph=False)
    model = StubScorer(data.num_nodes, data.num_relations)
    got = ueval.evaluate(model, data, batch_size=8, metrics=("mr", "mrr", "hits@10"))
    explicit = ueval.evaluate(model, data, batch_size=8, metrics=("mr", "mrr", "hits@10"), filtered_data=data.filtered_data)
    assert got == explicit
    train_only = Data(edge_index=data.edge_index, edge_type=data.edge_type, num_nodes=data.num_nodes, num_relations=data.num_relations)
    loose = ueval.evaluate(model, data, batch_size=8, metrics=("mr", "mrr", "hits@10"), filtered_data=train_only)
    assert loose["mr"] >= got["mr"]      # fewer known answers filtered out: ranks can only be worse
