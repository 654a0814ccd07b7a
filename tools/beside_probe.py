"""The entity layer in one launch, two forms (ultra_rspmm_forward_update): the update in the kernel's TAIL (update_form 1) vs
BESIDE the walk (update_form 3: twelve waves walk, four multiply the rows handed over through an LDS ring; round 4's form 2 --
rows by reference -- was removed in round 5 and prints as nan / "not served").  Prints times between
HIP events (back-to-back launches and inside a hipGraph of 20 launches), bit-equality with the two launches, and the
per-wave end-of-work clocks of one traced launch (walkers 0..11, update waves 12..15).

    python tools/beside_probe.py [shape] [batch] [sum]       env: ULTRA_STREAM_SHARES_12="q0,q1,q2" (calibration)
"""
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
grid = int(os.environ.get("PROBE_GRID", "256"))      # workgroups per launch (256: one per CU)


def agg_only():
    return plan.forward(rel, x, sum=agg_sum, mul="mul", point=point)


def two():
    return dense._conv_update_forward(x, agg_only(), w, b, lw, lb, 1e-5, 7)


EXTRA = int(os.environ.get("PROBE_FLAGS", "0"))     # 256: no matrix chain, 512: the update waves only drain their queue (wrong results)


def one(form):
    rspmm.set_tuning(update_form=form, grid=grid if grid != 256 else 0)
    out = plan.forward_update(rel, x, w, b, lw, lb, 1e-5, 7 | EXTRA, point=point, sum=agg_sum)
    rspmm.set_tuning()
    return out


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def graphed(fn, n=20, reps=10):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            gr.replay()
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


want = two()
served = False      # (form 2: removed)
served3 = one(3) is not None
nan = float("nan")
print("%s bs %d %s: tail == two launches %s | beside == two launches %s | beside through LDS == two launches %s" %
      (shape, bs, agg_sum, torch.equal(one(1), want), torch.equal(one(2), want) if served else "not served",
       torch.equal(one(3), want) if served3 else "not served"))
for k in range(2):
    print("back to back: aggregate %.1f us | two launches %.1f | tail %.1f | beside %.1f | beside through LDS %.1f" %
          (timed(agg_only), timed(two), timed(lambda: one(1)), timed(lambda: one(2)) if served else nan,
           timed(lambda: one(3)) if served3 else nan))
print("in a hipGraph of 20: aggregate %.1f us | two launches %.1f | tail %.1f | beside %.1f | beside through LDS %.1f" %
      (graphed(agg_only), graphed(two), graphed(lambda: one(1)), graphed(lambda: one(2)) if served else nan,
       graphed(lambda: one(3)) if served3 else nan))
# repeatability of the hand-off: 100 launches, every output compared
for form, ok in ((2, served), (3, served3)):
    if ok:
        outs = [one(form) for _ in range(100)]
        torch.cuda.synchronize()
        print("form %d, 100 launches all equal to the two launches: %s" % (form, all(torch.equal(o, want) for o in outs)))

if served3:
    for form in (1, 3):
        trace = torch.zeros(grid * 32, dtype=torch.int64, device=dev)
        one(form)
        torch.cuda.synchronize()
        _lib.check(_lib.lib.ultra_order_trace(trace.data_ptr()))
        one(form)
        torch.cuda.synchronize()
        _lib.check(_lib.lib.ultra_order_trace(None))
        t = trace.cpu()
        main = t[:3 * grid].view(grid, 3).double()
        wave_end = t[8 * grid:24 * grid].view(grid, 16).double() - main[:, :1]
        chain = main[:, 1] - main[:, 0]
        end = main[:, 2] - main[:, 0]
        print("form %d: cycles since the workgroup's start -- chains done %.0f, end mean %.0f max %.0f" %
              (form, chain.mean(), end.mean(), end.max()))
        late = end.argsort(descending=True)[:6].tolist()
        print("  the six workgroups that end last (end | chains done | walkers' last end): " +
              "  ".join("%.0f | %.0f | %.0f" % (end[i], chain[i], wave_end[i, :12].max()) for i in late))
        if form == 3:
            # workgroup b serves partition b // bs for sample b % bs: the same work eight times over
            per = end.view(-1, bs)
            pm = per.mean(dim=1)
            print("  per partition (mean over its %d samples): min %.0f max %.0f, std across partitions %.0f; std inside a partition %.0f (mean)"
                  % (bs, pm.min(), pm.max(), pm.std(), per.std(dim=1).mean()))
            order = pm.argsort(descending=True).tolist()
            wl = wave_end[:, :12].max(dim=1)[0].view(-1, bs).mean(dim=1)
            print("  partitions, last to first (end | chains done | walkers' last end): " +
                  "  ".join("%d: %.0f | %.0f | %.0f" % (q, pm[q], chain.view(-1, bs).mean(dim=1)[q], wl[q]) for q in order[:5] + order[-3:]))
        if form == 3 and os.environ.get("PROBE_DUMP_PARTS"):
            print("PARTS end " + " ".join("%.0f" % v for v in pm.tolist()))
            print("PARTS chain " + " ".join("%.0f" % v for v in chain.view(-1, bs).mean(dim=1).tolist()))
            print("PARTS walk " + " ".join("%.0f" % v for v in wl.tolist()))
        early = end.argsort()[:3].tolist()
        print("  the three that end first: " + "  ".join("%.0f | %.0f | %.0f" % (end[i], chain[i], wave_end[i, :12].max()) for i in early))
        print("  end of work per wave (mean over workgroups): " + " ".join("%6.0f" % v for v in wave_end.mean(dim=0).tolist()))
        print("  ... max over workgroups:                     " + " ".join("%6.0f" % v for v in wave_end.max(dim=0)[0].tolist()))
        if form == 3:
            phases = t[3 * grid:7 * grid].view(grid, 4).double().mean(dim=0).tolist()
            print("  update wave 0, cycles (mean over workgroups): waiting for a block %.0f | multiplying %.0f | until the pre-norm rows "
                  "are all written %.0f | finishing %.0f" % tuple(phases))
