"""The N > 1 path without a cluster: world_size-2 gloo processes on CPU.  The scorer is a small
deterministic stand-in module (the real Ultra needs a GPU); what is under test is the sharding, the
single all-gather, and that metrics equal the single-process result (script/run.py:121-226 protocol)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubScorer(torch.nn.Module):
    """score(h, t, r) = <emb[h] * rel[r], emb[t]>: any deterministic function of the batch works here."""

    def __init__(self, num_node, num_rel):
        super().__init__()
        g = torch.Generator().manual_seed(7)
        self.emb = torch.nn.Parameter(torch.randn(num_node, 8, generator=g))
        self.rel = torch.nn.Parameter(torch.randn(num_rel, 8, generator=g))

    def forward(self, data, batch):
        h, t, r = batch.unbind(-1)
        return (self.emb[h] * self.rel[r] * self.emb[t]).sum(-1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ultra_amd import distributed as udist
    from ultra_amd import eval as ueval
    from ultra_amd import synthetic
    data = synthetic.make_kg(num_node=60, num_triple=400, num_relation_base=3, num_test=37, seed=5, relation_graph=False)
    model = StubScorer(data.num_nodes, data.num_relations)
    res = ueval.evaluate(model, data, batch_size=4, metrics=("mr", "mrr", "hits@1", "hits@10", "hits@10_50", "mrr-tail"))
    # score-row all-gather (the per-step collective of the benchmark)
    score = torch.full((3, 5), float(rank))
    gathered = udist.all_gather_scores(score)
    var = udist.all_gather_variable(torch.arange(rank + 2) + 10 * rank)
    lo, hi = udist.shard_range(7)
    shards = udist.all_gather_shards(torch.arange(lo, hi).repeat_interleave(2).view(-1, 1) * torch.tensor([[1, 10]]), 7, rows_per_item=2)
    torch.save(dict(res=res, gathered=gathered, var=var, shards=shards, shard=udist.shard_range(37)),
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_query_sharded_evaluation_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from ultra_amd import eval as ueval
    from ultra_amd import synthetic
    data = synthetic.make_kg(num_node=60, num_triple=400, num_relation_base=3, num_test=37, seed=5, relation_graph=False)
    model = StubScorer(data.num_nodes, data.num_relations)
    want = ueval.evaluate(model, data, batch_size=4, metrics=("mr", "mrr", "hits@1", "hits@10", "hits@10_50", "mrr-tail"))
    outs = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    assert want["_num_rankings"] == 2 * 37
    for o in outs:
        assert o["res"]["_num_rankings"] == want["_num_rankings"]           # no padding, no duplicated samples
        for k, v in want.items():
            assert o["res"][k] == pytest.approx(v, rel=1e-6), k
        assert o["gathered"].shape == (6, 5)
        assert o["gathered"][:3].eq(0).all() and o["gathered"][3:].eq(1).all()   # rank-major
        assert o["var"].tolist() == [0, 1, 10, 11, 12]
        # one fixed-size collective for shard_range shares (4 + 3 items, two rows each): item order, no padding left
        assert o["shards"][:, 0].tolist() == [i for i in range(7) for _ in range(2)] and o["shards"][:, 1].tolist() == [10 * i for i in range(7) for _ in range(2)]
    assert outs[0]["shard"] == (0, 19) and outs[1]["shard"] == (19, 37)


WORLD8_KG = dict(num_node=48, num_triple=600, num_relation_base=4, num_test=20466, seed=9, relation_graph=False)
WORLD8_METRICS = ("mr", "mrr", "hits@1", "hits@3", "hits@10", "mrr-tail")


def _world8_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ultra_amd import distributed as udist
    from ultra_amd import eval as ueval
    from ultra_amd import synthetic
    data = synthetic.make_kg(**WORLD8_KG)
    model = StubScorer(data.num_nodes, data.num_relations)
    res = ueval.evaluate(model, data, batch_size=64, metrics=WORLD8_METRICS)
    lo, hi = udist.shard_range(data.target_triples.shape[0])
    # the per-step collective of the benchmark at this world size: every rank's (bs, N) score rows, rank-major
    gathered = udist.all_gather_scores(torch.full((2, 3), float(rank)))
    torch.save(dict(res=res, shard=(lo, hi), gathered=gathered), os.path.join(out_dir, "w%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_world_of_eight_ranks_with_uneven_shards_matches_single_process(tmp_path):
    """The 1 / 2 / 4 / 8-GPU half of BASELINE.json's metric has never met eight GPUs (no multi-GPU box: DESIGN.md section 7), so
    the eight-rank program is run on CPU: FB15k237's 20,466 test triples do not divide by 8 (six ranks of 2,558, two of 2,559 --
    script/run.py:127's DistributedSampler would pad by repeating samples), every rank evaluates its own shard through the real
    evaluate(), the ONE all-gather of the evaluation (distributed.all_gather_shards, run.py:165-186's six all-reduces) returns
    every ranking exactly once, and the metrics equal the single-process evaluation on every rank."""
    world = 8
    mp.spawn(_world8_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from ultra_amd import eval as ueval
    from ultra_amd import synthetic
    data = synthetic.make_kg(**WORLD8_KG)
    assert data.target_triples.shape[0] == 20466 and 20466 % world != 0
    want = ueval.evaluate(StubScorer(data.num_nodes, data.num_relations), data, batch_size=64, metrics=WORLD8_METRICS)
    assert want["_num_rankings"] == 2 * 20466
    outs = [torch.load(os.path.join(str(tmp_path), "w%d.pt" % r)) for r in range(world)]
    sizes = [o["shard"][1] - o["shard"][0] for o in outs]
    assert sorted(sizes) == [2558] * 6 + [2559] * 2 and sum(sizes) == 20466
    assert all(outs[r]["shard"][1] == outs[r + 1]["shard"][0] for r in range(world - 1))      # contiguous: no gap, no overlap
    for o in outs:
        assert o["res"]["_num_rankings"] == want["_num_rankings"]        # nothing padded, nothing counted twice
        for k, v in want.items():
            assert o["res"][k] == pytest.approx(v, rel=1e-6), k
        assert o["gathered"].shape == (16, 3)
        assert [float(o["gathered"][2 * r, 0]) for r in range(world)] == [float(r) for r in range(world)]


def _order_worker(rank, world, port, out_dir):
    """(a) rank 0's readout association reaches every rank; (b) the collectives a PipelinedForward issues from its
    alternating slots pair up across ranks in program order."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # the ranks start out DISAGREEING: rank 0 adds in a two-lane association read from a file, rank 1 sequentially
    os.environ["ULTRA_READOUT_ORDER"] = os.path.join(out_dir, "order.json") if rank == 0 else "sequential"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ultra_amd import distributed as udist
    from ultra_amd import graph as ugraph
    from ultra_amd import host_order
    before = host_order.order_id(host_order.readout_stages(128)[0])
    shared = udist.share_readout_order(128)
    after = host_order.order_id(host_order.readout_stages(128)[0])

    class Slot(object):          # a stub forward with its own output buffer, like a captured forward
        def __init__(self, k):
            self.k, self.out = k, torch.zeros(2, 3)

        def __call__(self, batch):
            self.out.copy_(batch * 100 + rank)
            return self.out

    made = []
    pf = ugraph.PipelinedForward(None, None, torch.zeros(2, 3), depth=2,
                                 slot_factory=lambda: made.append(Slot(len(made))) or made[-1])
    gathered = []
    for i in range(7):
        batch = torch.full((2, 3), float(i))
        gathered.append(pf(batch, post=lambda score: udist.all_gather_scores(score).clone()))
    pf.join()
    torch.save(dict(before=before, shared=shared, after=after, gathered=torch.stack(gathered), source=host_order.readout_stages(128)[1]),
               os.path.join(out_dir, "o%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_share_one_readout_order_and_pipelined_collectives_stay_in_program_order(tmp_path):
    sys.path.insert(0, ROOT)
    from ultra_amd import host_order
    two_lane = [(2, False, [list(range(0, 128, 2)), list(range(1, 128, 2))])]
    host_order.save_stages(str(tmp_path / "order.json"), two_lane, "two lanes (test)", 128)
    world = 2
    mp.spawn(_order_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "o%d.pt" % r)) for r in range(world)]
    want = host_order.order_id(two_lane)
    assert outs[0]["before"] == want and outs[1]["before"] == host_order.order_id(host_order.sequential_stages(128)) != want
    for o in outs:
        assert o["shared"] == want and o["after"] == want
    assert "broadcast" in outs[1]["source"]
    # step i's all-gather met step i's all-gather of the other rank, slots alternating: rows (rank 0's, rank 1's) = 100 i + rank
    for o in outs:
        for i in range(7):
            assert o["gathered"][i][:2].eq(100.0 * i).all() and o["gathered"][i][2:].eq(100.0 * i + 1).all()


def test_shard_range_is_a_partition():
    from ultra_amd import distributed as udist
    for n in (0, 1, 7, 8, 20466):
        for world in (1, 2, 3, 8):
            parts = [udist.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_metrics_formulas():
    from ultra_amd import eval as ueval
    ranking = torch.tensor([1, 2, 4, 10, 100])
    neg = torch.tensor([100, 100, 100, 100, 100])
    m = ueval.metrics_from_rankings(ranking, neg, ["mr", "mrr", "hits@1", "hits@3", "hits@10", "hits@1_50"])
    assert m["mr"] == pytest.approx(23.4)
    assert m["mrr"] == pytest.approx((1 + 0.5 + 0.25 + 0.1 + 0.01) / 5)
    assert m["hits@1"] == pytest.approx(0.2) and m["hits@3"] == pytest.approx(0.4) and m["hits@10"] == pytest.approx(0.8)
    fp = (ranking - 1).float() / neg
    assert m["hits@1_50"] == pytest.approx(float(((1 - fp) ** 49).mean()))
    with pytest.raises(ValueError):
        ueval.metrics_from_rankings(ranking, neg, ["auroc"])
