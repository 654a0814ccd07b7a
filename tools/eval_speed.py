"""Secondary number: the full filtered-ranking protocol (tail + head passes, filter, rank, metrics;
script/run.py:121-226) on the FB15k237-shaped graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import eval as ueval  # noqa: E402
from ultra_amd import models, synthetic  # noqa: E402

dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(dev)
model = models.Ultra(**synthetic.default_model_cfg())
model.load_state_dict(torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                              "ultra_3g_model.pt")))
model = model.to(dev).eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ueval.evaluate(model, data, batch_size=8, max_triples=16)
for cache in (False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ueval.evaluate(model, data, batch_size=8, max_triples=n, cache_relations=cache)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("evaluate(%d test triples, tail+head, filtered%s): %.3f s -> %.1f triples/s, %.2f M candidate scores/s; %s"
          % (n, ", relation representations of all relations computed once" if cache else "", dt, n / dt,
             2 * n * data.num_nodes / dt / 1e6, {k: round(v, 4) for k, v in res.items()}))
