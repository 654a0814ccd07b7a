"""Pins the model-level oracle (oracle/ultra_oracle_model.py) to the reference: golden scores recorded
from the unchanged reference modules + shipped checkpoints (tests/golden/gen_golden.py).  CPU only."""
import os

import pytest
import torch

from oracle import build_ref, ultra_oracle_model
from ultra_amd import synthetic
from ultra_amd.data import Data

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = [("ultra_3g", "sum"), ("ultra_50g", "max")]


def load_golden(ckpt, aggr):
    g = torch.load(os.path.join(GOLDEN, "model_%s_%s.pt" % (ckpt, aggr)))
    state = torch.load(os.path.join(GOLDEN, ckpt + "_model.pt"))
    data = Data(edge_index=g["edge_index"], edge_type=g["edge_type"], num_nodes=g["num_nodes"],
                num_relations=g["num_relations"],
                relation_graph=Data(edge_index=g["rel_edge_index"], edge_type=g["rel_edge_type"],
                                    num_nodes=g["num_relations"], num_relations=4))
    return g, state, data, synthetic.default_model_cfg(aggregate_func=aggr)


@pytest.mark.parametrize("ckpt,aggr", MODELS)
@pytest.mark.parametrize("kernel", ["c_oracle", "reference_tu"])
def test_oracle_model_matches_reference_scores(ckpt, aggr, kernel):
    g, state, data, cfg = load_golden(ckpt, aggr)
    fn = None
    if kernel == "reference_tu":
        if not build_ref.available():
            pytest.skip("oracle/_ref not built")
        fn = ultra_oracle_model.reference_rspmm_fn()
    for key_batch, key_pred in (("t_batch", "t_pred"), ("h_batch", "h_pred"), ("neg_batch", "neg_pred")):
        got = ultra_oracle_model.ultra_forward(state, cfg, data, g[key_batch], rspmm_fn=fn)
        assert got.shape == g[key_pred].shape
        # same algorithm, same dtype, different BLAS blocking / summation order only
        torch.testing.assert_close(got, g[key_pred], rtol=1e-4, atol=2e-5)


def test_head_batches_are_converted_to_tail_mode():
    """base_nbfnet.py:79-86: a head batch (t fixed) becomes a tail batch with the inverse relation."""
    h = torch.tensor([[5, 6, 7]])
    t = torch.tensor([[9, 9, 9]])
    r = torch.tensor([[2, 2, 2]])
    nh, nt, nr = ultra_oracle_model.negative_sample_to_tail(h, t, r, num_direct_rel=10)
    assert nh.tolist() == [[9, 9, 9]] and nt.tolist() == [[5, 6, 7]] and nr.tolist() == [[12, 12, 12]]
