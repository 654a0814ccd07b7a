"""Query sharding over the GPUs of a node.

Every query row (h, r, ?) is an independent Bellman-Ford propagation over a replicated graph, so the
path shards by queries with no data-path collective inside the forward; the reference does the same
with DistributedSampler (script/run.py:127) and then emulates an all-gather with six zero-padded
all_reduce(SUM) calls (script/run.py:166-186).  Here: one all-gather (RCCL over xGMI on GPUs, gloo on
CPU for the tests) of the per-rank score rows, or of the per-rank rankings.
"""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(num_items, rank_=None, world=None):
    """Contiguous, balanced [lo, hi) share of `num_items` for a rank (no padding, no duplicates,
    unlike DistributedSampler which repeats samples to equalise shard sizes)."""
    world = world_size() if world is None else world
    rank_ = rank() if rank_ is None else rank_
    base, extra = divmod(num_items, world)
    lo = rank_ * base + min(rank_, extra)
    return lo, lo + base + (1 if rank_ < extra else 0)


def all_gather_scores(score):
    """(b, N) per-rank score rows -> (world * b, N), rank-major.  One collective."""
    world = world_size()
    if not (dist.is_available() and dist.is_initialized()):
        return score
    score = score.contiguous()
    out = torch.empty((world * score.shape[0],) + tuple(score.shape[1:]), dtype=score.dtype, device=score.device)
    dist.all_gather_into_tensor(out, score)
    return out


def all_gather_shards(local, num_items, rows_per_item=1):
    """Rows of a quantity sharded by shard_range(num_items): rank r holds rows_per_item * (its share) rows of `local`
    (shape (rows, ...)); returns all rows in item order on every rank.  The shard sizes follow from num_items, so this is
    ONE collective: every rank pads to the largest share, all_gather_into_tensor, the padding is cut away."""
    world = world_size()
    if world == 1:
        return local
    shares = [shard_range(num_items, r, world) for r in range(world)]
    rows = [rows_per_item * (hi - lo) for lo, hi in shares]
    if local.shape[0] != rows[rank()]:
        raise ValueError("rank %d holds %d rows, its share of %d items is %d" % (rank(), local.shape[0], num_items, rows[rank()]))
    cap = max(rows)
    pad = local.new_zeros((cap,) + tuple(local.shape[1:]))
    pad[:local.shape[0]] = local
    out = local.new_empty((world * cap,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous())
    return torch.cat([out[r * cap:r * cap + rows[r]] for r in range(world)])


def all_gather_variable(t):
    """Concatenate 1-D tensors of different lengths from every rank (rank order).  Two collectives:
    lengths, then padded payloads."""
    world = world_size()
    if world == 1:
        return t
    n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.long)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s) for s in sizes]
    pad = torch.zeros(max(sizes), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])


def share_readout_order(K=128, device=None):
    """One association of the readout's last product for the whole job.  The order comes from the host BLAS
    (host_order.readout_stages: disk cache or a probe); left to themselves the ranks of a launcher each read / probe it --
    a cache race at best, two associations (different OMP / MKL environments per rank) at worst, and then the gathered
    score rows of one evaluation would not all be the same numbers.  Rank 0 resolves the order, its program is broadcast,
    every other rank adopts it; the ranks then compare order ids (one small all-gather) and raise on a mismatch.
    Returns the order id.  Call after init_process_group and before the first forward; device: where the collectives'
    tensors live (None: cuda for nccl, cpu for gloo)."""
    import json

    from . import host_order
    if world_size() == 1:
        return host_order.order_id(host_order.readout_stages(K)[0])
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    payload = b""
    if rank() == 0:
        stages, source = host_order.readout_stages(K)
        payload = json.dumps({"source": source, "stages": [[L, bool(carry), [list(map(int, lane)) for lane in lists]]
                                                           for L, carry, lists in stages]}).encode()
    n = torch.tensor([len(payload)], dtype=torch.long, device=device)
    dist.broadcast(n, 0)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=device)
    if rank() == 0:
        buf.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(buf, 0)
    if rank() != 0:
        rec = json.loads(bytes(buf.cpu().tolist()).decode())
        stages = [(int(L), bool(carry), [list(map(int, lane)) for lane in lists]) for L, carry, lists in rec["stages"]]
        host_order.adopt(stages, rec["source"] + " [rank 0's, broadcast]", K)
    mine = host_order.order_id(host_order.readout_stages(K)[0])
    word = torch.tensor([int(mine.split("-")[1], 16)], dtype=torch.long, device=device)
    words = [torch.zeros_like(word) for _ in range(world_size())]
    dist.all_gather(words, word)
    if any(int(w.item()) != int(word.item()) for w in words):
        raise RuntimeError("readout summation order differs across ranks: %s" % [hex(int(w.item())) for w in words])
    return mine
