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
