"""Per-workgroup clock trace of the reference-order kernel on the entity graph: chain phase and unit phase cycles per
workgroup beside the schedule's work (chunks, walk steps) -- calibrates plan.cpp's cost model.
    python tools/order_trace.py [shape] [bs] [chain_min]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import _lib, rspmm, synthetic  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "fb15k237"
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    chain_min = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dev = torch.device("cuda:0")
    data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
    N, R = data.num_nodes, data.num_relations
    g = torch.Generator().manual_seed(0)
    x = torch.randn(bs, N, 64, generator=g).to(dev)
    rel = torch.randn(bs, R, 64, generator=g).to(dev)
    point = (torch.arange(bs, device=dev) * 7 % N, torch.randn(bs, 64, generator=g).to(dev))
    plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True, seg_len=chain_min)
    grid = 256
    smod = min(bs, grid)
    nparts = grid // smod
    print(shape, "bs", bs, plan.info()["n_chain_row"], "chain rows;", plan.schedule_info(nparts))
    chunk_ptr, unit_ptr, units, chunks = plan.schedule(nparts)
    items = plan.export(_lib.ARR_ITEM).view(-1, 4)
    sdesc, _ = plan.streams(nparts)
    stream_steps = sdesc[:, 1].view(nparts, 64).sum(dim=1)
    n_chain = plan.info()["n_chain_row"]
    ms, _ = plan.forward_timed(rel, x, point=point, warmup=3, iters=20)
    print("time per call %.4f ms" % ms)
    trace = torch.zeros(grid * 32, dtype=torch.int64, device=dev)   # (see ultra_order_trace: 32 words per workgroup)
    _lib.check(_lib.lib.ultra_order_trace(trace.data_ptr()))
    plan.forward(rel, x, point=point)
    torch.cuda.synchronize()
    _lib.check(_lib.lib.ultra_order_trace(None))
    extra = trace[grid * 3:].cpu()
    t = trace[:grid * 3].cpu().view(grid, 3)
    t0 = t[:, 0].min()
    rows = []
    for b in range(grid):
        part = b // smod
        nch = int(chunk_ptr[part + 1] - chunk_ptr[part])
        ch_edges = int(chunks[chunk_ptr[part]:chunk_ptr[part + 1], 2].sum())
        ch_rows = int((chunks[chunk_ptr[part]:chunk_ptr[part + 1], 3] & 1).sum())
        us = units[unit_ptr[part]:unit_ptr[part + 1]].long()
        steps = int(items[n_chain + 4 * us, 2].sum()) if len(us) else 0
        rows.append((b, part, int(t[b, 0] - t0), int(t[b, 1] - t[b, 0]), int(t[b, 2] - t[b, 1]), nch, ch_edges, ch_rows, len(us), steps, int(stream_steps[part])))
    rows.sort(key=lambda r: -(r[3] + r[4]))
    print("block part start chain_cyc unit_cyc | chunks chain_edges chain_rows units unit_steps stream_steps")
    for r in rows[:12] + rows[-6:]:
        print("%5d %4d %6d %9d %8d | %6d %11d %10d %5d %10d %8d" % r)
    tot = torch.tensor([[r[3], r[4], r[5], r[6], r[7], r[8], r[9]] for r in rows], dtype=torch.float64)
    # least squares: chain_cyc ~ a * edges + b * chunks + c * rows ; unit_cyc ~ d * steps / 16 ... per workgroup
    A = tot[:, [3, 2, 4]]
    sol = torch.linalg.lstsq(A, tot[:, 0:1]).solution.flatten()
    print("chain fit: %.2f cyc/edge + %.1f cyc/chunk + %.1f cyc/row" % tuple(sol.tolist()))
    B = torch.stack([tot[:, 6], tot[:, 5], torch.ones(len(rows), dtype=torch.float64)], dim=1)
    sol2 = torch.linalg.lstsq(B, tot[:, 1:2]).solution.flatten()
    print("unit fit: %.2f cyc/step + %.1f cyc/unit + %.0f const   (clock = s_memtime ticks)" % tuple(sol2.tolist()))
    S = torch.stack([tot[:, 7], torch.ones(len(rows), dtype=torch.float64)], dim=1) if False else None
    st = torch.tensor([[r[10], 1.0] for r in rows], dtype=torch.float64)
    sol3 = torch.linalg.lstsq(st, tot[:, 1:2]).solution.flatten()
    print("stream fit (second phase vs group-stream steps): %.3f cyc/step + %.0f const" % tuple(sol3.tolist()))
    C = torch.tensor([[r[5], r[7], 1.0] for r in rows], dtype=torch.float64)
    sol4 = torch.linalg.lstsq(C, tot[:, 0:1]).solution.flatten()
    print("chain fit with constant: %.1f cyc/chunk + %.0f cyc/row + %.0f const" % tuple(sol4.tolist()))
    print("span of kernel: %d ticks; mean busy %d" % (int((t[:, 2] - t0).max()), int((t[:, 2] - t[:, 0]).double().mean())))


if __name__ == "__main__":
    main()
