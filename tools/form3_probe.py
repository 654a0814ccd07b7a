"""Form 3 of the one-launch entity layer (update beside the walk, rows through LDS), ONE line per run -- for sweeps over the
schedule's environment knobs (each needs its own process: the knobs are read when a schedule is built).

    python tools/form3_probe.py [shape] [batch] [sum]     env: PROBE_GRID (256), PROBE_DUMP_PARTS=1 (per-partition cycle dump),
                                                               ULTRA_STREAM_SHARES_12, ULTRA_CHAIN_LIMIT_FACTOR, ULTRA_STREAM_ROW_ORDER ...
(The PROBE_CALIBRATE loops of round 5 -- profile-guided partition budgets, then an incremental re-deal -- were removed with their
plan.cpp knobs after they showed no gain: profiles/r5_experiments.txt.)
Prints: us per layer inside a hipGraph of 20 (median of 5), bit-equality with the two launches, and from one traced launch the
cycles since each workgroup's start: chains done, walkers' last end, update waves' end (means over the workgroups), workgroup end
mean / max."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import _lib, dense, rspmm, synthetic  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "fb15k237"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
agg_sum = sys.argv[3] if len(sys.argv) > 3 else "add"
dev = torch.device("cuda:0")
data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
N, R = data.num_nodes, int(data.num_relations)
g = torch.Generator().manual_seed(0)
x = torch.randn(bs, N, 64, generator=g).to(dev)
rel = torch.randn(bs, R, 64, generator=g).to(dev)
point = (torch.arange(bs, device=dev) * 7 % N, torch.randn(bs, 64, generator=g).to(dev))
w = (torch.randn(64, 128, generator=g) / 11).to(dev)
b, lw, lb = (torch.randn(64, generator=g).to(dev) for _ in range(3))
plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
grid = int(os.environ.get("PROBE_GRID", "256"))
form = int(os.environ.get("PROBE_FORM", "3"))
flags = 7 | int(os.environ.get("PROBE_FLAGS", "0"))


def one():
    rspmm.set_tuning(update_form=form, grid=grid if grid != 256 else 0)
    out = plan.forward_update(rel, x, w, b, lw, lb, 1e-5, flags, point=point, sum=agg_sum)
    rspmm.set_tuning()
    return out


want = dense._conv_update_forward(x, plan.forward(rel, x, sum=agg_sum, mul="mul", point=point), w, b, lw, lb, 1e-5, 7)
got = one()
assert got is not None, "form %d does not serve this call" % form
equal = torch.equal(got, want)
torch.cuda.synchronize()
s = torch.cuda.Stream()
times = []
with torch.cuda.stream(s):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(20):
            one()
    gr.replay()
    torch.cuda.synchronize()
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            gr.replay()
        e1.record(s)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 100 * 1e3)
times.sort()
trace = torch.zeros(grid * 32, dtype=torch.int64, device=dev)
one()
torch.cuda.synchronize()
_lib.check(_lib.lib.ultra_order_trace(trace.data_ptr()))
one()
torch.cuda.synchronize()
_lib.check(_lib.lib.ultra_order_trace(None))
rspmm.check_device_error()
t = trace.cpu()
main = t[:3 * grid].view(grid, 3).double()
wave_end = t[8 * grid:24 * grid].view(grid, 16).double() - main[:, :1]
chain = main[:, 1] - main[:, 0]
end = main[:, 2] - main[:, 0]
retries = t[24 * grid:25 * grid].double()
print("%s bs %d %s grid %d form %d: %.2f us (min %.2f max %.2f) equal %s | chains %.0f walkers %.0f update %.0f | end mean %.0f max %.0f | ring-full retries mean %.1f max %.0f"
      % (shape, bs, agg_sum, grid, form, times[2], times[0], times[-1], equal, chain.mean(), wave_end[:, :12].max(dim=1)[0].mean(),
         wave_end[:, 12:].max(dim=1)[0].mean(), end.mean(), end.max(), retries.mean(), retries.max()))
if os.environ.get("PROBE_DUMP_PARTS"):
    nparts = grid // bs
    print("PARTS end " + " ".join("%.0f" % v for v in end.view(nparts, bs).mean(dim=1).tolist()))
    print("PARTS chain " + " ".join("%.0f" % v for v in chain.view(nparts, bs).mean(dim=1).tolist()))
    print("PARTS walk " + " ".join("%.0f" % v for v in wave_end[:, :12].max(dim=1)[0].view(nparts, bs).mean(dim=1).tolist()))
    print("PARTS upd " + " ".join("%.0f" % v for v in wave_end[:, 12:].max(dim=1)[0].view(nparts, bs).mean(dim=1).tolist()))
    for q in range(3):
        print("PARTS walkq%d " % q + " ".join("%.0f" % v for v in wave_end[:, 4 * q:4 * q + 4].max(dim=1)[0].view(nparts, bs).mean(dim=1).tolist()))
    for wv in range(16):      # end of every wave's work, mean over the partition's samples
        print("PARTS wave%d " % wv + " ".join("%.0f" % v for v in wave_end[:, wv].view(nparts, bs).mean(dim=1).tolist()))
