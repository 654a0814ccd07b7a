"""Per-partition cost model of form 3 from a traced launch: features of each partition's share of the twelve-walker schedule (CPU,
from the plan) against the measured cycles of tools/form3_probe.py PROBE_DUMP_PARTS=1.

    python tools/partition_fit.py <file with the probe's PARTS lines> [shape] [batch] [grid]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import rspmm, synthetic  # noqa: E402

path = sys.argv[1]
shape = sys.argv[2] if len(sys.argv) > 2 else "fb15k237"
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 256
nparts = grid // bs
parts = {}
want_grid = "grid %d" % grid
take = False
for line in open(path):
    if " form " in line and "grid" in line:
        take = want_grid in line and (shape + " bs") in line and (len(sys.argv) <= 5 or (" " + sys.argv[5] + " ") in line)
    if line.startswith("PARTS ") and take:
        name, vals = line.split()[1], np.array([float(v) for v in line.split()[2:]])
        if len(vals) == nparts:
            parts[name] = vals
data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
N, R = data.num_nodes, int(data.num_relations)
plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
key = nparts | (1 << 24)
sdesc, srec = plan.streams(nparts, walkers=12)
sdesc = sdesc.view(nparts, 64, 2)
srec = srec.numpy()
deg_src = np.bincount(data.edge_index[1].numpy(), minlength=N)      # how often a node is gathered: hot rows hit the caches
steps = sdesc[:, :, 1].sum(dim=1).numpy().astype(float)
maxlen = sdesc[:, :, 1].max(dim=1)[0].numpy().astype(float)
rows = np.zeros(nparts)
hot = np.zeros(nparts)      # gathers of the 350 hottest source rows
cold = np.zeros(nparts)
order = np.argsort(-deg_src)
hotset = np.zeros(N, dtype=bool)
hotset[order[:1000]] = True
for q in range(nparts):
    for g in range(48):
        first, n = int(sdesc[q, g, 0]), int(sdesc[q, g, 1])
        seg = srec[first:first + n]
        mk = seg[:, 1] == R
        rows[q] += mk.sum()
        cols = seg[~mk, 0]
        hot[q] += hotset[cols].sum()
        cold[q] += (~hotset[cols]).sum()
n = ctypes = None
import ctypes  # noqa: E402
from ultra_amd._lib import check, lib  # noqa: E402


def sched_array(which):
    cnt = ctypes.c_int64()
    check(lib.ultra_plan_schedule_export(plan._h, key, which, None, 0, ctypes.byref(cnt)))
    t = torch.empty(cnt.value, dtype=torch.int32)
    check(lib.ultra_plan_schedule_export(plan._h, key, which, t.data_ptr(), cnt.value, ctypes.byref(cnt)))
    return t


cp = sched_array(0).numpy()
chunks = (cp[1:] - cp[:-1]).astype(float)
print("partitions %d: steps %.0f..%.0f rows %.0f..%.0f chunks %.0f..%.0f" % (nparts, steps.min(), steps.max(), rows.min(), rows.max(), chunks.min(), chunks.max()))
for target in ("walk", "upd", "end"):
    if target not in parts:
        continue
    y = parts[target] - (parts["chain"] if target != "end" else 0)
    for names, cols in ((("steps",), [steps]), (("steps", "rows"), [steps, rows]), (("hot", "cold", "rows"), [hot, cold, rows]),
                        (("hot", "cold", "rows", "chunks"), [hot, cold, rows, chunks]), (("steps", "rows", "chunks", "maxlen"), [steps, rows, chunks, maxlen])):
        A = np.stack(cols + [np.ones(nparts)], axis=1)
        coef, res, _, _ = np.linalg.lstsq(A, y, rcond=None)
        pred = A @ coef
        print("%-5s ~ %-32s coef %s  rms %.0f  (std of target %.0f)" % (target, " + ".join(names), " ".join("%.2f" % c for c in coef), np.sqrt(((y - pred) ** 2).mean()), y.std()))
worst = np.argsort(-parts["end"])[:6]
print("last partitions: " + "  ".join("%d: end %.0f steps %.0f rows %.0f chunks %.0f hot %.0f" % (q, parts["end"][q], steps[q], rows[q], chunks[q], hot[q]) for q in worst))
best = np.argsort(parts["end"])[:4]
print("first partitions: " + "  ".join("%d: end %.0f steps %.0f rows %.0f chunks %.0f hot %.0f" % (q, parts["end"][q], steps[q], rows[q], chunks[q], hot[q]) for q in best))
