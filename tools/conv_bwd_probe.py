"""ultra_conv_update_backward (rows kernel + weights kernel + reduce) per call, HIP events over a hipGraph of 20 calls, for the row
counts of the fine-tuning step.  ULTRA_CONV_BWD_SHAPE=blocks_cap,wpb_rows,wpb_weights selects the launch shape (0 = the library's)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ultra_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")


def run(rows, iters=20):
    g = torch.Generator().manual_seed(0)
    x, agg, gout = (torch.randn(rows, 64, generator=g).to(dev) for _ in range(3))
    w = (torch.randn(64, 128, generator=g) / 11).to(dev)
    b, lw, lb = (torch.randn(64, generator=g).to(dev) for _ in range(3))
    gx, gagg, gw = torch.empty_like(x), torch.empty_like(x), torch.empty_like(w)
    gb, glw, glb = torch.empty_like(b), torch.empty_like(b), torch.empty_like(b)
    nbytes = lib.ultra_conv_update_backward_workspace(rows)
    work = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)

    def call():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.ultra_conv_update_backward(x.data_ptr(), agg.data_ptr(), gout.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(),
                                             lb.data_ptr(), gx.data_ptr(), gagg.data_ptr(), gw.data_ptr(), gb.data_ptr(), glw.data_ptr(),
                                             glb.data_ptr(), work.data_ptr(), nbytes, rows, 64, 64, 1e-5, 7 | int(os.environ.get("PROBE_FLAGS", "0")), st))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            call()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    if int(os.environ.get("PROBE_FLAGS", "0")) & 2048:      # CB_DBG_CLOCK: workgroup 0's shader-clock and 100-MHz ticks of the last call
        ticks = work[:4].view(torch.int64).tolist()
        print("  %d rows: workgroup 0 ran %d shader cycles in %.1f us = %.0f MHz" % (rows, ticks[0], ticks[1] / 100.0,
                                                                                 ticks[0] / (ticks[1] / 100.0)), flush=True)
    return e0.elapsed_time(e1) / (5 * iters) * 1e3, float(gw.abs().sum())


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [2056, 3792, 14541 * 8, 123182 * 8]
    print("shape", os.environ.get("ULTRA_CONV_BWD_SHAPE", "auto"),
          "  ".join("%d rows %.1f us (|gw| %.4f)" % ((r,) + run(r)) for r in sizes), flush=True)
