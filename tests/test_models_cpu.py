"""Host logic of the module API without a GPU: checkpoint drop-in (state-dict keys), configuration
plumbing, and loud failure on CPU tensors (no silent fallback)."""
import os

import pytest
import torch

from tests.test_oracle_model import GOLDEN, load_golden
from ultra_amd import models, synthetic


@pytest.mark.parametrize("ckpt", ["ultra_3g", "ultra_50g"])
def test_reference_checkpoints_load_unchanged(ckpt):
    state = torch.load(os.path.join(GOLDEN, ckpt + "_model.pt"))
    model = models.Ultra(**synthetic.default_model_cfg())
    missing, unexpected = model.load_state_dict(state, strict=True)
    assert not missing and not unexpected
    assert sum(p.numel() for p in model.parameters()) == 168705     # README.md:98
    assert set(model.state_dict().keys()) == set(state.keys())


def test_cfg_dicts_are_not_consumed():
    cfg = synthetic.default_model_cfg()
    models.Ultra(**cfg)
    assert cfg["rel_model_cfg"]["class"] == "RelNBFNet" and cfg["entity_model_cfg"]["class"] == "EntityNBFNet"


def test_cpu_tensors_fail_loudly():
    g, state, data, cfg = load_golden("ultra_3g", "sum")
    model = models.Ultra(**cfg).eval()
    model.load_state_dict(state)
    with pytest.raises(RuntimeError, match="no CPU path"):
        with torch.no_grad():
            model(data, g["t_batch"])


def test_negative_sample_to_tail():
    net = models.BaseNBFNet(64, [64], 1)
    h = torch.tensor([[5, 6, 7], [1, 1, 1]])
    t = torch.tensor([[9, 9, 9], [2, 3, 4]])
    r = torch.tensor([[2, 2, 2], [0, 0, 0]])
    nh, nt, nr = net.negative_sample_to_tail(h, t, r, num_direct_rel=10)
    assert nh.tolist() == [[9, 9, 9], [1, 1, 1]] and nt.tolist() == [[5, 6, 7], [2, 3, 4]]
    assert nr.tolist() == [[12, 12, 12], [0, 0, 0]]


def test_point_boundary_is_the_reference_boundary_tensor():
    """layers.PointBoundary.dense() == zeros + scatter_add_ of one row per sample (models.py:59-66, 135-141)."""
    from ultra_amd import layers
    g = torch.Generator().manual_seed(3)
    bs, n, d = 5, 17, 64
    rows = torch.randint(0, n, (bs,), generator=g)
    vals = torch.randn(bs, d, generator=g)
    want = torch.zeros(bs, n, d)
    want.scatter_add_(1, rows.view(bs, 1, 1).expand(-1, 1, d), vals.unsqueeze(1))
    point = layers.PointBoundary(rows[::1], vals, n)
    assert torch.equal(point.dense(), want)
    assert point.rows.is_contiguous() and point.rows.dtype == torch.long and not point.requires_grad
    # strided row ids (batch[:, 0, 2] in Ultra.forward) are made contiguous once
    strided = torch.stack([rows, rows + 1], dim=1)[:, 0]
    assert layers.PointBoundary(strided, vals, n).rows.is_contiguous()


def test_fast_path_switches_default_on():
    """The inference fast paths are module-level switches (A/B tests flip them); they must ship enabled."""
    from ultra_amd import layers, models
    assert layers.ONEHOT_FAST_PATH and layers.POINT_BOUNDARY_FAST_PATH and layers.FUSED_DENSE_LAYER
    assert models.PROLOGUE_FAST_PATH


def test_relation_projection_has_the_reference_interface():
    """ultra/ultraquery.py:245-277: RelationProjection(model, threshold=0.0).forward(graph, h_prob, r_index)."""
    import inspect

    from ultra_amd import models, synthetic
    from ultra_amd.ultraquery import RelationProjection
    cfg = synthetic.default_model_cfg()
    cfg["entity_model_cfg"]["class"] = "QueryNBFNet"
    model = models.Ultra(**cfg)
    proj = RelationProjection(model, threshold=0.25)
    assert proj.model is model and proj.threshold == 0.25
    assert list(inspect.signature(RelationProjection.__init__).parameters) == ["self", "model", "threshold"]
    assert list(inspect.signature(proj.forward).parameters) == ["graph", "h_prob", "r_index"]
    assert {k for k in proj.state_dict()} == {"model." + k for k in model.state_dict()}


def test_raw_triples_reader_builds_the_reference_fact_graph(tmp_path):
    """ultra_amd.data.load_triples_dir: training triples + inverses as the fact graph, test triples as targets
    (ultra/datasets.py:186-197), with and without vocabulary files."""
    from ultra_amd.data import load_triples_dir
    (tmp_path / "train.txt").write_text("a\tlikes\tb\nb\tlikes\tc\nc\tknows\ta\n")
    (tmp_path / "valid.txt").write_text("a\tknows\tc\n")
    (tmp_path / "test.txt").write_text("b\tknows\ta\nc\tlikes\tb\n")
    d = load_triples_dir(str(tmp_path), relation_graph=False)
    assert d.num_nodes == 3 and d.num_relations == 4 and d.num_edges == 6
    assert d.edge_index.tolist() == [[0, 1, 2, 1, 2, 0], [1, 2, 0, 0, 1, 2]]
    assert d.edge_type.tolist() == [0, 0, 1, 2, 2, 3]
    assert d.target_triples.tolist() == [[1, 0, 1], [2, 1, 0]]
    (tmp_path / "entities.dict").write_text("0\tc\n1\tb\n2\ta\n")
    (tmp_path / "relations.dict").write_text("0\tknows\n1\tlikes\n")
    d2 = load_triples_dir(str(tmp_path), relation_graph=False)
    assert d2.edge_index[:, :3].tolist() == [[2, 1, 0], [1, 0, 2]] and d2.edge_type[:3].tolist() == [1, 1, 0]
    d3 = load_triples_dir(str(tmp_path))          # with the relation graph (tasks.build_relation_graph)
    assert d3.relation_graph.num_nodes == 4


