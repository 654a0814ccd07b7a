"""Two batches in flight: two captured forwards (own input / output / activation buffers each) replayed alternately on two
streams against the same replays on one stream.  Measures whether independent batches fill the launches that leave the chip
idle (relation model, glue) -- DESIGN.md section 8, item 0.    python tools/two_stream_probe.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ultra_amd import graph as ugraph, synthetic, tasks  # noqa: E402
import secondary_bench as sb  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bs = 8
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(dev)
model = sb.load_model("sum", "ultra_3g").eval()
batches = [tasks.all_negative(data, data.target_triples[i * bs:(i + 1) * bs])[0] for i in range(4)]
fwd = [ugraph.GraphedForward(model, data, batches[0]) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
want = [fwd[0](b).clone() for b in batches]
torch.cuda.synchronize()


def one_stream():
    for i in range(steps):
        fwd[i % 2](batches[i % 4])


def two_streams():
    for i in range(steps):
        with torch.cuda.stream(streams[i % 2]):
            fwd[i % 2](batches[i % 4])


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for s in streams:
    s.wait_stream(torch.cuda.current_stream())
print("one stream  %.4f ms per step" % timed(one_stream))
print("two streams %.4f ms per step" % timed(two_streams))
# the last two replays of two_streams(): steps - 2 and steps - 1
got = [fwd[(steps - 2) % 2].static_out, fwd[(steps - 1) % 2].static_out]
ok = [bool(torch.equal(got[k], want[(steps - 2 + k) % 4])) for k in range(2)]
print("scores of the last two concurrent replays equal the sequential ones:", ok)
print("one stream  %.4f ms per step" % timed(one_stream))
print("two streams %.4f ms per step" % timed(two_streams))
