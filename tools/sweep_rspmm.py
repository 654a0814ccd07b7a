"""GPU micro-benchmark sweep of the rspmm forward kernel over plan / launch parameters.

    python tools/sweep_rspmm.py [--shape fb15k237] [--bs 8] [--out gpurun_out/sweep.json]

Prints HIP-event kernel time, algorithmic (gather-model) GB/s and the fraction of the 8 TB/s HBM peak
for the entity graph and its relation graph (B_gather = 4 D (E + N + R) + 12 E + 4 (N + 1)).
"""
import argparse
import itertools
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import rspmm, synthetic  # noqa: E402


def b_gather(E, N, R, D, boundary=False):
    return 4 * D * (E + N + R + (N if boundary else 0)) + 12 * E + 4 * (N + 1)


def run(plan_args, feats, sum, seg_len, g_max, tuning, iters, boundary=None):
    ei, et, N, R = plan_args
    rel, x = feats
    plan = rspmm.Plan(ei, et, N, R, seg_len=seg_len, g_max=g_max)
    rspmm.set_tuning(**tuning)
    ms, _ = plan.forward_timed(rel, x, boundary=boundary, sum=sum, warmup=3, iters=iters)
    rspmm.set_tuning()
    info = plan.info()
    del plan
    return ms, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="fb15k237")
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="gpurun_out/sweep.json")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    data = synthetic.make_kg(**synthetic.SHAPES[args.shape], seed=1234)
    D = args.bs * 64
    g = torch.Generator().manual_seed(0)
    results = []
    graphs = {
        "entity": (data.edge_index, data.edge_type, data.num_nodes, data.num_relations),
        "relation": (data.relation_graph.edge_index, data.relation_graph.edge_type, data.relation_graph.num_nodes, 4),
    }
    for gname, (ei, et, N, R) in graphs.items():
        E = ei.shape[1]
        # batch-major operands (bs, rows, 64): the module-level layout
        x = torch.randn(args.bs, N, 64, generator=g).to(dev)
        rel = torch.randn(args.bs, R, 64, generator=g).to(dev)
        bnd = torch.randn(args.bs, N, 64, generator=g).to(dev)
        bytes_alg = b_gather(E, N, R, D)
        segs = [256] if args.quick else [128, 256, 512, 1024]
        gmaxs = [32] if args.quick else [4, 16, 32, 64, 128]
        tunings = [dict()] if args.quick else [dict(), dict(threads=512), dict(threads=256), dict(rel_lds=0, x_lds=0),
                                                dict(x_lds=0), dict(grid=512), dict(grid=512, threads=512)]
        for sum in (["add"] if args.quick else ["add", "max"]):
            for seg, gm, tn in itertools.product(segs, gmaxs, tunings):
                if gm > seg:
                    continue
                if sum == "max" and (seg != 256 or tn):
                    continue
                try:
                    ms, info = run((ei, et, N, R), (rel, x), sum, seg, gm, tn, args.iters)
                except Exception as exc:  # keep sweeping
                    print("FAIL", gname, sum, seg, gm, tn, exc, flush=True)
                    continue
                gbs = bytes_alg / (ms * 1e-3) / 1e9
                rec = dict(graph=gname, sum=sum, seg_len=seg, g_max=gm, tuning=tn, ms=ms, gbs=gbs, frac=gbs / 8000.0,
                           n_unit=info["n_unit"], n_wave_item=info["n_wave_item"], n_group_item=info["n_group_item"],
                           n_slot=info["n_partial_slot"], E=E, N=N, R=R, D=D)
                results.append(rec)
                print("%-8s %-3s seg=%-4d gmax=%-3d %-32s %8.3f ms %8.1f GB/s  %5.1f%% of 8TB/s  units=%d" %
                      (gname, sum, seg, gm, json.dumps(tn), ms, gbs, 100 * gbs / 8000.0, info["n_unit"]), flush=True)
        # fused boundary epilogue and node-major layout at the default plan
        ms, _ = run((ei, et, N, R), (rel, x), "add", 0, 0, dict(), args.iters, boundary=bnd)
        print("%-8s add default plan + fused boundary: %.3f ms" % (gname, ms), flush=True)
        results.append(dict(graph=gname, sum="add", variant="default+boundary", ms=ms))
        x2 = x.transpose(0, 1).reshape(N, D).contiguous()
        rel2 = rel.transpose(0, 1).reshape(R, D).contiguous()
        ms, _ = run((ei, et, N, R), (rel2, x2), "add", 0, 0, dict(), args.iters)
        print("%-8s add default plan, node-major (N, D) layout: %.3f ms" % (gname, ms), flush=True)
        results.append(dict(graph=gname, sum="add", variant="default node-major", ms=ms))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)
    best = {}
    for r in results:
        if "gbs" in r and r["sum"] == "add":
            k = r["graph"]
            if k not in best or r["ms"] < best[k]["ms"]:
                best[k] = r
    print("BEST", json.dumps(best))


if __name__ == "__main__":
    main()
