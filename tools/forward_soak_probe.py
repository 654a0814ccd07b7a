"""A long run of the pipelined forward (three captured graphs in flight, as the benchmark and evaluate() run it) against the same
batches scored one at a time: every score of every batch compared BIT FOR BIT.  The hand-off between the walking and the multiplying
waves of the entity layer and the sharing of CUs between the batches in flight are exercised N times with N different batches.
Usage: python tools/forward_soak_probe.py [batches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import secondary_bench as sb  # noqa: E402
from ultra_amd import rspmm, synthetic, tasks  # noqa: E402
from ultra_amd.graph import GraphedForward, PipelinedForward  # noqa: E402

n_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(sb.dev)
model = sb.load_model("sum", "ultra_3g").eval()
triples = data.target_triples
with torch.no_grad():
    def batch(i):
        lo = (8 * i) % (len(triples) - 8)
        t, h = tasks.all_negative(data, triples[lo:lo + 8])
        return h if i % 3 == 2 else t                   # (head batches too)
    single = GraphedForward(model, data, batch(0))
    want = [single(batch(i)).clone() for i in range(n_batch)]
    torch.cuda.synchronize()
    piped = PipelinedForward(model, data, batch(0), depth=3)
    got = [piped(batch(i), post=lambda s: s.clone()) for i in range(n_batch)]
    piped.join()
    torch.cuda.synchronize()
    rspmm.check_device_error()
bad = [i for i in range(n_batch) if not torch.equal(got[i], want[i])]
print("%d batches of 8 x %d candidates (a third of them head batches): %d differ between three in flight and one at a time%s" % (
    n_batch, data.num_nodes, len(bad), "" if not bad else " -- first: %s" % bad[:5]), flush=True)
