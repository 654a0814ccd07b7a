"""readout timing (8 x 14541 all-tail candidates) with the host-BLAS order program and with the sequential chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import dense, host_order, models, synthetic

dev = torch.device("cuda:0")
bs, n = 8, 14541
net = models.EntityNBFNet(**{k: v for k, v in synthetic.default_model_cfg()["entity_model_cfg"].items() if k != "class"}).to(dev)
g = torch.Generator().manual_seed(0)
hidden = torch.randn(bs, n, 64, generator=g).to(dev)
query = torch.randn(bs, 64, generator=g).to(dev)
t_index = torch.arange(n).unsqueeze(0).expand(bs, -1).contiguous().to(dev)


def run(order, iters=50):
    for _ in range(5):
        dense.readout(net, hidden, query, t_index, order=order)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dense.readout(net, hidden, query, t_index, order=order)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


stages, source = host_order.readout_stages(128)
print(source, [(L, c, [len(l) for l in ls]) for L, c, ls in stages])
print("host order   %.1f us" % run(dense.readout_order(dev)))
seq = torch.tensor(host_order.stages_to_program(host_order.sequential_stages(128)), dtype=torch.int32, device=dev)
print("sequential   %.1f us" % run(seq))
