"""Parity at the sizes BASELINE.json's configs name, on the path bench.py times (default plans = reference summation
order): Ultra.forward on the GPU against oracle/ultra_oracle_model.py with the reference's own rspmm.cpp TU
(oracle/_ref) -- torch CPU ops in the reference's data flow -- over several all-tail / all-head batches each.

    config 1  ultra_3g,  WN18RR shape,   batch 4   (sum aggregate; the reference's CPU-runnable plumbing case)
    config 2  ultra_3g,  FB15k237 shape, batch 8   (distmult + sum: the headline)
    config 3  ultra_50g, CoDEx-L shape,  batch 8   (max aggregate)
    config 5  ultra_50g, YAGO3-10 shape, batch 8   (sum aggregate; the forward of the fine-tuning configuration)

Gates: north_star asks scores within 1e-4 and identical rankings.  Every operation of the forward follows the reference's
order (rspmm.cpp's sequential row sums, torch's nn.Linear / nn.LayerNorm arithmetic, the host BLAS's association for the
readout's last product -- ultra_amd/host_order.py), so where that association was recovered from this host's BLAS the
test asserts what is then true: >= 99.9 % of the scores BIT-EQUAL (the BLAS may sum a few trailing rows of each thread's
share with a remainder kernel), max difference <= 1e-5, and strictly identical rankings.  On a host whose BLAS tree is
outside the recognised family (sequential fallback) the gates are 1e-5 and rankings identical up to the reference's own
near-ties: a GPU rank may differ only if moving the positive's REFERENCE score by <= 2 * max|gpu - reference| reproduces it."""
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
    ("config5_yago310_forward", "ultra_50g", "yago310", 8, "sum", 1),       # the fine-tuning config's graph, forward only
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
    # (ULTRA_PARITY_BATCHES=n: that many batches instead -- one-off long runs, profiles/r6_experiments.txt)
    n_batch = int(os.environ.get("ULTRA_PARITY_BATCHES", n_batch))
    data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234)
    cfg = synthetic.default_model_cfg(aggregate_func=aggr)
    state = torch.load(os.path.join(GOLDEN, ckpt + "_model.pt"))
    model = models.Ultra(**cfg)
    model.load_state_dict(state)          # strict: the checkpoint's keys are the module's keys
    model = model.to(dev).eval()
    gdata = data.to(dev)
    fn = ultra_oracle_model.reference_rspmm_fn()
    worst_diff, strict, outside, total, equal, count = 0.0, 0, 0, 0, 0, 0
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
            equal += int((got == want).sum())
            count += got.numel()
            r_got = tasks.compute_ranking(got, pos, mask)
            r_want = tasks.compute_ranking(want, pos, mask)
            best, worst = _tie_band(want, pos, mask, 2 * diff)
            strict += int((r_got != r_want).sum())
            outside += int(((r_got < best) | (r_got > worst)).sum())
            total += len(pos)
    msg = "%s: max |gpu - reference| = %.3g, %.4f %% of the scores bit-equal, %d of %d rankings differ, %d outside the " \
          "reference's ties" % (name, worst_diff, 100.0 * equal / count, strict, total, outside)
    print(msg)
    assert worst_diff <= 1e-5, msg
    assert outside == 0, msg
    from ultra_amd import host_order
    if aggr == "max" or host_order.readout_stages(128)[1].startswith("host BLAS"):
        assert strict == 0, msg
    if host_order.readout_stages(128)[1].startswith("host BLAS"):
        assert equal >= 0.999 * count, msg


def test_hits_at_k_on_top_ranked_queries(dev):
    """north_star: "identical Hits@k rankings".  The benchmark's synthetic test triples rank near 3,000 of 14,541 (Hits@k = 0 on
    both sides: vacuous), so this batch takes 16 fact-graph edges whose reference ranks are 1 .. 30
    (tests/golden/topk_queries_fb15k237.json, gen_topk_queries.py): MRR and Hits@1/3/10 of the GPU path must equal the
    reference flow's to 6 digits, on non-zero values (script/run.py:188-213, ultra/tasks.py:133-141)."""
    import json
    rec = json.load(open(os.path.join(GOLDEN, "topk_queries_fb15k237.json")))
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234)
    cfg = synthetic.default_model_cfg()
    state = torch.load(os.path.join(GOLDEN, "ultra_3g_model.pt"))
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()
    gdata = data.to(dev)
    fn = ultra_oracle_model.reference_rspmm_fn()
    triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)
    queries = triples[torch.tensor(rec["indices"])]
    r_gpu, r_ref = [], []
    for b in range(0, len(queries), 8):
        batch = queries[b:b + 8]
        cand, _ = tasks.all_negative(data, batch)
        mask, _ = tasks.strict_negative_mask(data, batch)
        want = ultra_oracle_model.ultra_forward(state, cfg, data, cand, rspmm_fn=fn)
        with torch.no_grad():
            got = model(gdata, cand.to(dev)).cpu()
        r_gpu.append(tasks.compute_ranking(got, batch[:, 1], mask))
        r_ref.append(tasks.compute_ranking(want, batch[:, 1], mask))
    r_gpu, r_ref = torch.cat(r_gpu), torch.cat(r_ref)
    assert r_ref.tolist() == rec["reference_ranks"]          # the fixture's ranks reproduce on this host

    def metrics(r):
        r = r.double()
        return [round((1 / r).mean().item(), 6)] + [round((r <= k).double().mean().item(), 6) for k in (1, 3, 10)]
    m_gpu, m_ref = metrics(r_gpu), metrics(r_ref)
    print("top-k fixture: MRR / Hits@1 / Hits@3 / Hits@10  gpu %s  reference %s" % (m_gpu, m_ref))
    assert 0 < m_ref[1] < m_ref[2] < m_ref[3] < 1
    assert m_gpu == m_ref and torch.equal(r_gpu, r_ref)
