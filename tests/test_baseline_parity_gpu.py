"""Parity at the sizes BASELINE.json's configs name, on the path bench.py times (default plans = reference summation
order): Ultra.forward on the GPU against oracle/ultra_oracle_model.py with the reference's own rspmm.cpp TU
(oracle/_ref) -- torch CPU ops in the reference's data flow -- over several all-tail / all-head batches each.

    config 1  ultra_3g,  WN18RR shape,   batch 4   (sum aggregate; the reference's CPU-runnable plumbing case)
    config 2  ultra_3g,  FB15k237 shape, batch 8   (distmult + sum: the headline)
    config 3  ultra_50g, CoDEx-L shape,  batch 8   (max aggregate)

Gates: scores within 1e-5 (north_star asks 1e-4; the hidden states are bit-equal, only the readout's last 128 -> 1
product -- MKL GEMV order on the CPU side -- differs); rankings from tasks.compute_ranking identical, where "identical"
tolerates exactly the reference's own near-ties: a GPU rank may differ only if moving the positive's REFERENCE score by
<= 2 * max|gpu - reference| reproduces it (on these synthetic graphs most positives sit in the bulk of unreachable
nodes whose scores agree to the last bits).  Strict equality is asserted for the max-aggregate config, whose scores
are spread out."""
import os

import pytest
import torch

from oracle import ultra_oracle_model
from ultra_amd import models, synthetic, tasks

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

CONFIGS = [
    # name, checkpoint, shape, batch size, aggregate, batches (each scored as tail AND head batch)
    ("config1_wn18rr", "ultra_3g", "wn18rr", 4, "sum", 4),
    ("config2_fb15k237", "ultra_3g", "fb15k237", 8, "sum", 4),
    ("config3_codex_l", "ultra_50g", "codex_l", 8, "max", 2),
]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _tie_band(ref_score, pos, mask, band):
    idx = torch.arange(len(pos))
    shifted = ref_score.clone()
    shifted[idx, pos] = ref_score[idx, pos] + band
    best = tasks.compute_ranking(shifted, pos, mask)
    shifted[idx, pos] = ref_score[idx, pos] - band
    worst = tasks.compute_ranking(shifted, pos, mask)
    return best, worst


@pytest.mark.parametrize("name,ckpt,shape,bs,aggr,n_batch", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_scores_and_rankings_at_baseline_size(dev, name, ckpt, shape, bs, aggr, n_batch):
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234)
    cfg = synthetic.default_model_cfg(aggregate_func=aggr)
    state = torch.load(os.path.join(GOLDEN, ckpt + "_model.pt"))
    model = models.Ultra(**cfg)
    model.load_state_dict(state)          # strict: the checkpoint's keys are the module's keys
    model = model.to(dev).eval()
    gdata = data.to(dev)
    fn = ultra_oracle_model.reference_rspmm_fn()
    worst_diff, strict, outside, total = 0.0, 0, 0, 0
    for b in range(n_batch):
        batch = data.target_triples[b * bs:(b + 1) * bs]
        t_batch, h_batch = tasks.all_negative(data, batch)
        t_mask, h_mask = tasks.strict_negative_mask(data, batch)
        pos_h, pos_t, _ = batch.t()
        for cand, pos, mask in ((t_batch, pos_t, t_mask), (h_batch, pos_h, h_mask)):
            want = ultra_oracle_model.ultra_forward(state, cfg, data, cand, rspmm_fn=fn)
            with torch.no_grad():
                got = model(gdata, cand.to(dev)).cpu()
            diff = (got - want).abs().max().item()
            worst_diff = max(worst_diff, diff)
            r_got = tasks.compute_ranking(got, pos, mask)
            r_want = tasks.compute_ranking(want, pos, mask)
            best, worst = _tie_band(want, pos, mask, 2 * diff)
            strict += int((r_got != r_want).sum())
            outside += int(((r_got < best) | (r_got > worst)).sum())
            total += len(pos)
    msg = "%s: max |gpu - reference| = %.3g, %d of %d rankings differ, %d outside the reference's ties" % (
        name, worst_diff, strict, total, outside)
    print(msg)
    assert worst_diff <= 1e-5, msg
    assert outside == 0, msg
    if aggr == "max":
        assert strict == 0, msg
