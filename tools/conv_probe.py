"""conv_update timing next to its matrix and byte floors."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import dense, _lib
from ultra_amd._lib import lib, check

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 14541 * 8
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, 64, generator=g).to(dev)
agg = torch.randn(rows, 64, generator=g).to(dev)
w = torch.randn(64, 128, generator=g).to(dev) / 11
b = torch.randn(64, generator=g).to(dev)
lw = torch.randn(64, generator=g).to(dev)
lb = torch.randn(64, generator=g).to(dev)
out = torch.empty_like(x)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(flags, iters=50):
    for _ in range(5):
        check(lib.ultra_conv_update(x.data_ptr(), agg.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                    out.data_ptr(), rows, 64, 64, 1e-5, flags, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        check(lib.ultra_conv_update(x.data_ptr(), agg.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                    out.data_ptr(), rows, 64, 64, 1e-5, flags, st))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("rows", rows)
print("full (LN+relu+residual)      %.1f us" % run(7))
print("no LN                        %.1f us" % run(6))
print("no matrix chain (measurement)  %.1f us" % run(7 | 256))
print("no matrix chain, no LN          %.1f us" % run(6 | 256))
flops = rows * 128 * 64 * 2
print("matrix floor @157 TF         %.1f us;  bytes floor @5 TB/s %.1f us" % (flops / 157e12 * 1e6, rows * 64 * 4 * 3 / 5e12 * 1e6))
