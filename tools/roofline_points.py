"""rspmm forward at the benchmark point and at HBM-bound points (working set > 256 MB Infinity Cache),
plus a streaming-copy ceiling.  Run plain for HIP-event timings, or under
    rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d <dir> -- python tools/roofline_points.py --pmc
    rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d <dir> -- python tools/roofline_points.py --pmc
(separate passes; --pmc mode launches each kernel a fixed small number of times, in a fixed order)."""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import _lib, rspmm, synthetic  # noqa: E402

POINTS = [("fb15k237", 8), ("fb15k237", 64), ("codex_l", 8), ("yago310", 8)]
COPY_BYTES = 1 << 30


def b_gather(E, N, R, D, boundary):
    return 4 * D * (E + N + R + (N if boundary else 0)) + 12 * E + 4 * (N + 1)


def b_min(E, N, R, D, boundary):
    return 4 * D * (2 * N + R + (N if boundary else 0)) + 12 * E + 4 * (N + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--out", default="gpurun_out/roofline_points.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {"points": []}
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    src = torch.empty(COPY_BYTES // 4, device=dev).normal_()
    dst = torch.empty_like(src)
    n_copy = 3 if args.pmc else 20
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.check(_lib.lib.ultra_stream_copy(dst.data_ptr(), src.data_ptr(), COPY_BYTES, stream))
    e0.record()
    for _ in range(n_copy):
        _lib.check(_lib.lib.ultra_stream_copy(dst.data_ptr(), src.data_ptr(), COPY_BYTES, stream))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n_copy
    res["copy"] = {"bytes_read": COPY_BYTES, "bytes_written": COPY_BYTES, "ms": ms,
                   "gbs_read_plus_write": 2 * COPY_BYTES / (ms * 1e-3) / 1e9, "launches": n_copy + 1}
    print("copy 1 GiB: %.3f ms -> %.0f GB/s (read + write)" % (ms, res["copy"]["gbs_read_plus_write"]), flush=True)
    del src, dst
    for shape, bs in POINTS:
        data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
        N, R, E, D = data.num_nodes, data.num_relations, data.num_edges, bs * 64
        g = torch.Generator().manual_seed(0)
        x = torch.randn(bs, N, 64, generator=g).to(dev)
        rel = torch.randn(bs, R, 64, generator=g).to(dev)
        bnd = torch.randn(bs, N, 64, generator=g).to(dev)
        point = (torch.arange(bs, device=dev) * 7 % N, torch.randn(bs, 64, generator=g).to(dev))
        plan = rspmm.Plan(data.edge_index, data.edge_type, N, R)
        for sum in ("add", "max"):
            iters = 3 if args.pmc else 20
            # sum: the boundary condition is one row per sample (ultra_rspmm_forward_point, what the forward passes);
            # max: a full boundary tensor (zero is not the identity of max)
            kw = dict(point=point) if sum == "add" else dict(boundary=bnd)
            ms, _ = plan.forward_timed(rel, x, sum=sum, mul="mul", warmup=1 if args.pmc else 3, iters=iters, **kw)
            alg, low = b_gather(E, N, R, D, sum != "add"), b_min(E, N, R, D, sum != "add")
            rec = dict(shape=shape, bs=bs, sum=sum, N=N, E=E, R=R, D=D, ms=ms, launches=iters + (1 if args.pmc else 3),
                       b_gather=alg, b_min=low, gbs_gather=alg / (ms * 1e-3) / 1e9, gbs_min=low / (ms * 1e-3) / 1e9,
                       x_plus_out_MB=2 * 4 * D * N / 1e6, info={k: v for k, v in plan.info().items() if k.startswith("n_")})
            res["points"].append(rec)
            print("%-9s bs=%-3d %s  %.3f ms  gather %.0f GB/s (%.0f%% of 8 TB/s)  compulsory %.0f GB/s  x+out %.0f MB" %
                  (shape, bs, sum, ms, rec["gbs_gather"], rec["gbs_gather"] / 80, rec["gbs_min"], rec["x_plus_out_MB"]),
                  flush=True)
        del plan, x, rel, bnd
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
