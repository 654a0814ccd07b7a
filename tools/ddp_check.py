"""One data-parallel fine-tuning step (DistributedDataParallel over RCCL) on a small synthetic KG; run with
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/ddp_check.py
Checks that gradients are finite and identical on every rank after the all-reduce."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from ultra_amd import models, synthetic, tasks

rank, local = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(0)
data = synthetic.make_kg(num_node=800, num_triple=8000, num_relation_base=6, num_test=16, seed=5).to(dev)
model = models.Ultra(**synthetic.default_model_cfg()).to(dev).train()
ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local])
opt = torch.optim.AdamW(ddp.parameters(), lr=5e-4)
triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)
batch = triples[rank * 4:(rank + 1) * 4]
neg = tasks.negative_sampling(data, batch, 16, strict=True)
pred = ddp(data, neg)
target = torch.zeros_like(pred)
target[:, 0] = 1
loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target)
loss.backward()
g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
assert torch.isfinite(g).all()
ref = g.clone()
dist.broadcast(ref, 0)
assert torch.equal(ref, g), "gradients differ across ranks after the all-reduce"
opt.step()
if rank == 0:
    print("ddp step ok: loss %.4f, %d gradient values, world %d" % (loss.item(), g.numel(), dist.get_world_size()))
dist.destroy_process_group()
