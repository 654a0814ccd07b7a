"""Per-WAVE cost model of the stream walk of form 3 from a traced launch (tools/form3_probe.py PROBE_DUMP_PARTS=1): for every walker
wave its end of walk (mean over the partition's samples) minus the partition's chain phase, against the steps and rows of its four
streams in the twelve-walker schedule.  One fit per wave quartet (the CU issues its oldest waves first: waves 0-3, 4-7, 8-11 step
at different rates).

    python tools/wave_fit.py <probe output> [shape] [batch] [grid]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import rspmm, synthetic  # noqa: E402

path = sys.argv[1]
shape = sys.argv[2] if len(sys.argv) > 2 else "fb15k237"
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 256
nparts = grid // bs
parts, take = {}, False
for line in open(path):
    if " form " in line and "grid" in line:
        take = ("grid %d" % grid) in line and shape in line and "wave11" not in parts
    if line.startswith("PARTS ") and take:
        vals = np.array([float(v) for v in line.split()[2:]])
        if len(vals) == nparts:
            parts[line.split()[1]] = vals
data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
N, R = data.num_nodes, int(data.num_relations)
plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
sdesc, srec = plan.streams(nparts, walkers=12)
sdesc = sdesc.view(nparts, 64, 2).numpy()
srec = srec.numpy()
mk = (srec[:, 1] == R).astype(np.int64)
cum = np.concatenate([[0], np.cumsum(mk)])
rows = np.zeros((nparts, 64))
for q in range(nparts):
    for g in range(48):
        first, n = int(sdesc[q, g, 0]), int(sdesc[q, g, 1])
        rows[q, g] = cum[first + n] - cum[first]
steps = sdesc[:, :, 1].astype(float)
S = steps[:, :48].reshape(nparts, 12, 4).max(axis=2)        # the wave walks its longest stream's steps
Ssum = steps[:, :48].reshape(nparts, 12, 4).sum(axis=2)
Rw = rows[:, :48].reshape(nparts, 12, 4).sum(axis=2)        # every flush of one of its four streams stalls the wave
T = np.stack([parts["wave%d" % w] for w in range(12)], axis=1) - parts["chain"][:, None]
print("%s bs %d, %d partitions: wave steps %.0f..%.0f, wave rows %.0f..%.0f" % (shape, bs, nparts, S.min(), S.max(), Rw.min(), Rw.max()))
for q in range(3):
    s_, r_, t_ = S[:, 4 * q:4 * q + 4].ravel(), Rw[:, 4 * q:4 * q + 4].ravel(), T[:, 4 * q:4 * q + 4].ravel()
    for names, cols in ((("steps",), [s_]), (("steps", "rows"), [s_, r_])):
        A = np.stack(cols + [np.ones_like(s_)], axis=1)
        coef = np.linalg.lstsq(A, t_, rcond=None)[0]
        print("quartet %d: walk ~ %-12s coef %s  rms %.0f (std %.0f)" % (q, " + ".join(names), " ".join("%.1f" % c for c in coef),
                                                                         np.sqrt(((t_ - A @ coef) ** 2).mean()), t_.std()))
    # without an intercept: what a schedule can use (time = c steps + r rows)
    A = np.stack([s_, r_], axis=1)
    coef = np.linalg.lstsq(A, t_, rcond=None)[0]
    print("quartet %d: walk ~ c steps + r rows (no constant): c %.1f r %.1f  -> a row costs %.1f steps; rms %.0f" %
          (q, coef[0], coef[1], coef[1] / coef[0], np.sqrt(((t_ - A @ coef) ** 2).mean())))
# all quartets at once: c_q steps + r rows (one row cost)
A = np.zeros((nparts * 12, 4))
t_all = T.ravel()
for q in range(3):
    sel = np.zeros((nparts, 12), dtype=bool)
    sel[:, 4 * q:4 * q + 4] = True
    A[sel.ravel(), q] = S[sel]
A[:, 3] = Rw.ravel()
coef = np.linalg.lstsq(A, t_all, rcond=None)[0]
print("all waves: c0 %.1f c1 %.1f c2 %.1f cycles a step, r %.1f cycles a row; rms %.0f" % (coef[0], coef[1], coef[2], coef[3], np.sqrt(((t_all - A @ coef) ** 2).mean())))

# ---- gathers by how hot their source row is (how often the graph gathers it: hot rows hit the vector L1 / L2) ----
deg_src = np.bincount(data.edge_index[1].numpy(), minlength=N)
rank = np.empty(N, dtype=np.int64)
rank[np.argsort(-deg_src)] = np.arange(N)
for cut in (300, 1000, 3000):
    cold = np.zeros((nparts, 64))
    for q in range(nparts):
        for g in range(48):
            first, n = int(sdesc[q, g, 0]), int(sdesc[q, g, 1])
            seg = srec[first:first + n]
            cols = seg[seg[:, 1] != R, 0]
            cold[q, g] = (rank[cols] >= cut).sum()
    Cw = cold[:, :48].reshape(nparts, 12, 4).max(axis=2)       # (the wave's slowest stream)
    Csum = cold[:, :48].reshape(nparts, 12, 4).sum(axis=2)
    A = np.zeros((nparts * 12, 7))
    for q in range(3):
        sel = np.zeros((nparts, 12), dtype=bool)
        sel[:, 4 * q:4 * q + 4] = True
        A[sel.ravel(), q] = S[sel]
        A[sel.ravel(), 3 + q] = Csum[sel] / 4.0
    A[:, 6] = Rw.ravel()
    coef = np.linalg.lstsq(A, t_all, rcond=None)[0]
    print("cold = source rank >= %d: steps c0 %.0f c1 %.0f c2 %.0f | extra per cold gather %.0f %.0f %.0f | row %.0f | rms %.0f"
          % (cut, coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], np.sqrt(((t_all - A @ coef) ** 2).mean())))
# per quartet: share of cold gathers and mean row length of its streams
for q in range(3):
    sl = slice(16 * q, 16 * q + 16)
    print("quartet %d: steps a stream %.0f, rows a stream %.1f, cold share (rank >= 1000) %.3f" %
          (q, steps[:, sl].mean(), rows[:, sl].mean(), 0.0 if steps[:, sl].sum() == 0 else
           sum((rank[srec[int(sdesc[p_, g, 0]):int(sdesc[p_, g, 0]) + int(sdesc[p_, g, 1])][:, 0][srec[int(sdesc[p_, g, 0]):int(sdesc[p_, g, 0]) + int(sdesc[p_, g, 1])][:, 1] != R]] >= 1000).sum()
               for p_ in range(nparts) for g in range(16 * q, 16 * q + 16)) / max(1.0, (steps[:, sl] - rows[:, sl]).sum())))
