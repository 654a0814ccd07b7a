"""Queries whose positives rank in the top 10: the fixture behind the non-trivial Hits@k parity check
(tests/test_baseline_parity_gpu.py::test_hits_at_k_on_top_ranked_queries, bench.py's parity block).

The synthetic test triples of the benchmark rank around 3,000 of 14,541, so Hits@1/3/10 of both sides are 0 and their
equality says nothing.  Edges of the fact graph itself are often recovered in the top 10.  This script scores the first
training triples of the FB15k237-shaped graph (seed 1234, ultra_3g weights) with the CPU oracle (reference rspmm.cpp TU when
oracle/_ref is built) and records 16 of them -- ranks 1, 2..3 and 4..10 mixed, so that Hits@1 < Hits@3 < Hits@10 < 1 is
possible -- as indices into the training triples.  Run from the repo root:  python tests/golden/gen_topk_queries.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ultra_oracle_model  # noqa: E402
from ultra_amd import synthetic, tasks  # noqa: E402


def main():
    data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234)
    cfg = synthetic.default_model_cfg()
    state = torch.load(os.path.join(ROOT, "tests", "golden", "ultra_3g_model.pt"))
    fn = ultra_oracle_model.reference_rspmm_fn()
    triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)[: data.num_edges // 2]
    found = {}
    for start in range(0, 1024, 8):
        batch = triples[start:start + 8]
        cand, _ = tasks.all_negative(data, batch)
        mask, _ = tasks.strict_negative_mask(data, batch)
        score = ultra_oracle_model.ultra_forward(state, cfg, data, cand, rspmm_fn=fn)
        rank = tasks.compute_ranking(score, batch[:, 1], mask)
        for i, r in enumerate(rank.tolist()):
            if r <= 40:
                found[start + i] = r
        buckets = [sum(1 for r in found.values() if lo <= r <= hi) for lo, hi in ((1, 1), (2, 3), (4, 10), (11, 40))]
        print(start, buckets, flush=True)
        if min(buckets) >= 4:
            break
    picks = []
    for lo, hi in ((1, 1), (2, 3), (4, 10), (11, 40)):
        picks += sorted(k for k, r in found.items() if lo <= r <= hi)[:4]
    picks.sort()
    out = {"shape": "fb15k237", "seed": 1234, "weights": "ultra_3g", "source": "training triples (fact-graph edges), tail mode",
           "indices": picks, "reference_ranks": [found[k] for k in picks]}
    with open(os.path.join(ROOT, "tests", "golden", "topk_queries_fb15k237.json"), "w") as f:
        json.dump(out, f)
    print(out)


if __name__ == "__main__":
    main()
