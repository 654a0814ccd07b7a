"""How the multi-rank paths are STARTED (reference: README.md:247-254 torch.distributed.launch, ultra/util.py:121-122 NCCL
init, script/run.py:44-45 DDP, run.py:127 sharding) -- exercised at world size 1 on the one GPU of the test box:

  * `bench.py` under torch.distributed.run: RCCL communicator + hipGraph replay + the per-step all-gather, and the
    per-rank probe; `bench.py --gpus N` with fewer than N GPUs visible must refuse, not report one GPU as N;
  * one DistributedDataParallel step over RCCL gives the gradients of the plain step.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def test_bench_under_torchrun_world_size_one():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--no-roofline", "--no-secondary"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]          # exactly one JSON line on stdout (RCCL's banner goes to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["value"] > 0
    cfg = out["config"]
    assert cfg["rccl_world_size"] == 1 and cfg["launch"].startswith("hipGraph")
    assert cfg["per_rank"]["probe_scores_identical"] is True and len(cfg["per_rank"]["ms_per_step"]) == 1
    assert cfg["readout_order_id"].startswith("order-")
    # one association of the readout's last product for the whole job (distributed.share_readout_order)
    assert cfg["per_rank"]["readout_order_identical"] is True and cfg["per_rank"]["readout_order_id"] == [cfg["readout_order_id"]]
    # the timed loop is repeated in the process: the first run is the line's figure, the others show the spread
    rep = out["repeats"]
    assert rep["runs"] == 5 and len(rep["ms_per_step"]) == 5 and rep["min"] <= rep["median"] <= rep["max"]
    assert rep["ms_per_step"][0] == pytest.approx(out["ms_per_step"], rel=1e-3)


def test_bench_scores_real_triples_from_files():
    """bench.py --data-root on the committed kg-datasets-layout fixture (tests/golden/kg_fixture): `data: real`, and the
    parity block against the oracle flow on the same batch is green."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--data-root", os.path.join(ROOT, "tests", "golden", "kg_fixture"),
           "--steps", "5", "--warmup", "2", "--repeats", "2", "--no-roofline", "--no-secondary", "--cpu-seconds", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["data"] == "real" and "kg_fixture" in out["config"]["workload"] and "N=300" in out["config"]["workload"]
    par = out["parity"]
    assert par["rank_mismatches"] == 0 and par["max_abs_score_diff"] <= 1e-5
    assert par["scores_bit_equal"] >= 0.999 * par["scores"]
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a launcher whose world size contradicts --gpus is refused as well
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not match" in r.stderr


DDP_STEP = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from ultra_amd import layers, models, synthetic, tasks
# (the plain step and the DDP step are compared bit for bit below, on the DEFAULT route: since round 6 the last layer's backward
# on the candidates' rows is a gather in a fixed order too -- ultra_rspmm_rows_backward_gather -- so every sum of the step is)
assert layers.LAST_LAYER_ON_ROWS
rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm (ultra/util.py:121-122)
data = synthetic.make_kg(num_node=800, num_triple=8000, num_relation_base=6, num_test=16, seed=5).to(dev)
triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)
torch.manual_seed(3)
neg = tasks.negative_sampling(data, triples[rank * 4:(rank + 1) * 4], 16, strict=True)

def grads(wrap):
    torch.manual_seed(0)
    model = models.Ultra(**synthetic.default_model_cfg()).to(dev).train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local]) if wrap else model   # script/run.py:44-45
    pred = net(data, neg)
    target = torch.zeros_like(pred)
    target[:, 0] = 1
    torch.nn.functional.binary_cross_entropy_with_logits(pred, target).backward()
    return torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])

plain, ddp = grads(False), grads(True)
assert torch.isfinite(ddp).all() and ddp.numel() > 1000
ref = ddp.clone()
dist.broadcast(ref, 0)
assert torch.equal(ref, ddp), "gradients differ across ranks after the all-reduce"
# world size 1: the all-reduce averages one contribution, so the DDP step IS the plain step
assert dist.get_world_size() > 1 or torch.equal(plain, ddp), (plain - ddp).abs().max().item()
print("DDP_OK world %%d grads %%d" %% (dist.get_world_size(), ddp.numel()))
dist.destroy_process_group()
"""


def test_ddp_step_over_rccl_equals_the_plain_step(tmp_path):
    script = tmp_path / "ddp_step.py"
    script.write_text(DDP_STEP % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DDP_OK world 1" in r.stdout, (r.stdout + r.stderr)[-2000:]


DDP_TWO_RANKS = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from ultra_amd import models, synthetic, tasks
rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0)                            # both ranks on the one GPU of the box: RCCL refuses that, gloo does not
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
data = synthetic.make_kg(num_node=800, num_triple=8000, num_relation_base=6, num_test=16, seed=5).to(dev)
triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)
torch.manual_seed(3 + rank)
neg = tasks.negative_sampling(data, triples[rank * 4:(rank + 1) * 4], 16, strict=True)      # every rank its own shard (run.py:33)

def grads(wrap):
    torch.manual_seed(0)
    model = models.Ultra(**synthetic.default_model_cfg()).to(dev).train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0]) if wrap else model   # script/run.py:44-45
    pred = net(data, neg)
    target = torch.zeros_like(pred)
    target[:, 0] = 1
    torch.nn.functional.binary_cross_entropy_with_logits(pred, target).backward()
    return torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])

plain, ddp = grads(False), grads(True)
parts = [torch.empty_like(plain) for _ in range(2)]
dist.all_gather(parts, plain)
mean = (parts[0] + parts[1]) / 2
assert not torch.equal(parts[0], parts[1]), "the two ranks must see different batches"
err = (ddp - mean).abs().max().item()
assert err <= 1e-6 * max(1.0, mean.abs().max().item()), err      # the all-reduce averaged the two ranks' gradients
ref = ddp.clone()
dist.broadcast(ref, 0)
assert torch.equal(ref, ddp), "gradients differ across ranks after the all-reduce"
print("DDP2_OK rank %%d grads %%d err %%.3g" %% (rank, ddp.numel(), err))
dist.destroy_process_group()
"""


def test_ddp_step_with_two_ranks_averages_their_gradients(tmp_path):
    """BASELINE config 5 is data parallel (script/run.py:44-45).  One GPU per test box and RCCL refuses two ranks on one device,
    so the TWO-rank step runs over gloo with both ranks on cuda:0: every rank a different batch, the DDP gradients must be the
    mean of the two plain steps' gradients and identical on both ranks -- the engine's autograd nodes under DDP's bucketed
    all-reduce hooks, at a world size where the all-reduce is not the identity."""
    script = tmp_path / "ddp_two.py"
    script.write_text(DDP_TWO_RANKS % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("DDP2_OK") == 2, (r.stdout + r.stderr)[-3000:]


def test_bench_with_two_ranks_sharing_the_gpu_over_gloo():
    """The N = 2 code path of bench.py on real forwards: two ranks (both on the box's one GPU, collectives over gloo -- RCCL
    refuses two ranks on one device), every step's score all-gather issued behind the forward of its pipeline slot, rank 0's
    readout association shared, per-rank clocks and probe digests gathered.  Not a measurement: the line says so."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6",
           "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-roofline", "--no-secondary"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["rccl_world_size"] == 2 and "gloo" in cfg["collective_backend"]
    per = cfg["per_rank"]
    assert per["probe_scores_identical"] is True and per["readout_order_identical"] is True and len(per["ms_per_step"]) == 2
    assert out["value"] == pytest.approx(2 * 8 * 14541 * 6 / (out["ms_per_step"] * 6e-3), rel=1e-6)      # whole-job aggregate


GRAPHED_TWO_RANKS = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from ultra_amd import models, synthetic, tasks, train
rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0)                            # both ranks on the one GPU of the box: gloo carries the all-reduce
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
data = synthetic.make_kg(num_node=800, num_triple=8000, num_relation_base=6, num_test=16, seed=5).to(dev)
triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)
torch.manual_seed(3 + rank)
batches = [tasks.negative_sampling(data, triples[(2 * i + rank) * 4:(2 * i + rank + 1) * 4], 16, strict=True) for i in range(3)]

def fresh():
    torch.manual_seed(0)
    return models.Ultra(**synthetic.default_model_cfg()).to(dev).train()

# the reference's way (script/run.py:44-45, 63-82): DistributedDataParallel around the model, AdamW, three steps
model = fresh()
net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
opt = train.make_adamw(model, lr=5e-3)
want_loss = [train.train_step(net, data, b, opt, num_negative=16).item() for b in batches]
want = [p.detach().clone() for p in model.parameters()]

# the captured step: forward + backward as one graph, ONE all-reduce of the flat gradient bucket, AdamW as a second graph
model = fresh()
opt = train.make_adamw(model, lr=5e-3, capturable=True)
step = train.GraphedTrainStep(model, data, opt, batches[0], num_negative=16)
assert step.world == 2 and step.step_graph is not None
got_loss = [step(b).item() for b in batches]
for a, b in zip(got_loss, want_loss):
    assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (got_loss, want_loss)
worst = 0.0
for p, q in zip(model.parameters(), want):
    worst = max(worst, (p - q).abs().max().item() / max(1.0, q.abs().max().item()))
assert worst <= 1e-5, worst                              # (DDP averages per bucket, the captured step one flat bucket: same mean to rounding)
flat = torch.cat([p.detach().flatten() for p in model.parameters()])
ref = flat.clone()
dist.broadcast(ref, 0)
assert torch.equal(ref, flat), "parameters differ across ranks after three captured steps"
print("GRAPHED2_OK rank %%d worst %%.3g" %% (rank, worst))
dist.destroy_process_group()
"""


def test_captured_training_step_with_two_ranks_matches_ddp(tmp_path):
    """train.GraphedTrainStep at world size 2 (gloo, both ranks on cuda:0): the same losses and parameters after three steps as
    DistributedDataParallel + train_step on the same per-rank batches, and identical parameters on both ranks."""
    script = tmp_path / "graphed_two.py"
    script.write_text(GRAPHED_TWO_RANKS % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("GRAPHED2_OK") == 2, (r.stdout + r.stderr)[-3000:]
