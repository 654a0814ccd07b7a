"""Evaluation glue on the GPU: fused filtered-rank kernel == tasks.compute_ranking over strict_negative_mask,
and the full evaluate() protocol equals a plain restatement of script/run.py:121-226 on the same scores."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("n,bs", [(60, 5), (5000, 8), (14541, 8)])
def test_filtered_rank_kernel_matches_masked_ranking(dev, n, bs):
    from ultra_amd import synthetic, tasks
    data = synthetic.make_kg(num_node=n, num_triple=8 * n, num_relation_base=3, num_test=64, seed=n).to(dev)
    batch = torch.stack([data.edge_index[0, :bs], data.edge_index[1, :bs], data.edge_type[:bs]], dim=-1)   # true triples
    g = torch.Generator().manual_seed(1)
    for quantised in (False, True):
        pred = torch.randn(bs, n, generator=g)
        if quantised:
            pred = (pred * 2).round() / 2          # many exact ties: they count against the positive (tasks.py:137)
        pred = pred.to(dev)
        t_mask, h_mask = tasks.strict_negative_mask(data, batch)
        for mode, mask, pos in (("tail", t_mask, batch[:, 1]), ("head", h_mask, batch[:, 0])):
            want = tasks.compute_ranking(pred, pos, mask)
            rank, num_neg = tasks.filtered_ranking(data, batch, pred, mode=mode)
            assert torch.equal(rank, want)
            assert torch.equal(num_neg, mask.sum(dim=-1))
    with pytest.raises(RuntimeError, match="no CPU path"):
        tasks.filtered_ranking(data.to("cpu"), batch.cpu(), pred.cpu())


def test_evaluate_matches_reference_protocol(dev):
    from tests.test_oracle_model import load_golden
    from ultra_amd import eval as ueval
    from ultra_amd import models, synthetic, tasks
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=400, num_triple=3000, num_relation_base=5, num_test=21, seed=3).to(dev)
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()
    res = ueval.evaluate(model, data, batch_size=8, metrics=("mr", "mrr", "hits@1", "hits@10", "mrr-tail"))
    # restatement of script/run.py:131-150 + 188-224 with the mask-based ranking
    triples = torch.cat([data.target_edge_index, data.target_edge_type.unsqueeze(0)]).t()
    ranks, tails = [], []
    with torch.no_grad():
        for s in range(0, len(triples), 8):
            batch = triples[s:s + 8]
            t_batch, h_batch = tasks.all_negative(data, batch)
            t_mask, h_mask = tasks.strict_negative_mask(data, batch)
            t_rank = tasks.compute_ranking(model(data, t_batch), batch[:, 1], t_mask)
            h_rank = tasks.compute_ranking(model(data, h_batch), batch[:, 0], h_mask)
            ranks += [t_rank, h_rank]
            tails += [t_rank]
    ranking, tail = torch.cat(ranks).float(), torch.cat(tails).float()
    assert res["_num_rankings"] == 42
    assert res["mr"] == pytest.approx(ranking.mean().item())
    assert res["mrr"] == pytest.approx((1 / ranking).mean().item())
    assert res["hits@1"] == pytest.approx((ranking <= 1).float().mean().item())
    assert res["hits@10"] == pytest.approx((ranking <= 10).float().mean().item())
    assert res["mrr-tail"] == pytest.approx((1 / tail).mean().item())


def test_cached_relation_representations_give_the_same_bits(dev):
    """Ultra.cache_relation_representations: the relation model's output depends on the query relation only (models.py:20-21),
    so a table of all of them, computed once, serves every batch -- same scores bit for bit, same metrics from evaluate();
    a table whose weights have moved is dropped, not used."""
    from tests.test_oracle_model import load_golden
    from ultra_amd import eval as ueval
    from ultra_amd import models, synthetic, tasks
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=400, num_triple=3000, num_relation_base=5, num_test=40, seed=3).to(dev)
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()
    t_batch, h_batch = tasks.all_negative(data, data.target_triples[:8])
    with torch.no_grad():
        plain_t, plain_h = model(data, t_batch), model(data, h_batch)
        table = model.cache_relation_representations(data)
        assert tuple(table.shape) == (data.num_relations, data.num_relations, 64)
        assert torch.equal(model(data, t_batch), plain_t) and torch.equal(model(data, h_batch), plain_h)
        # stale table: a weight of the relation model changes -> the next forward recomputes (and equals an uncached run)
        model.relation_model.layers[0].linear.bias.add_(0.25)
        moved = model(data, t_batch)
        assert getattr(model, "_rel_table", None) is None and not torch.equal(moved, plain_t)
        model.relation_model.layers[0].linear.bias.sub_(0.25)
    names = ("mr", "mrr", "hits@1", "hits@10", "mrr-tail")
    with_table = ueval.evaluate(model, data, batch_size=8, metrics=names, cache_relations=True)
    without = ueval.evaluate(model, data, batch_size=8, metrics=names, cache_relations=False)
    assert with_table == without and getattr(model, "_rel_table", None) is None


def test_graph_replay_stays_correct_when_interleaved_with_eager_work(dev):
    """Regression: hipGraph replays of the forward interleaved with eager forwards / ranking kernels must keep
    matching the eager scores (memset NODES captured from hipMemsetAsync were observed to go stale on ROCm 7.2
    once eager memsets interleave; the captured path uses fill kernels instead)."""
    from tests.test_oracle_model import load_golden
    from ultra_amd import eval as ueval
    from ultra_amd import models, synthetic, tasks
    from ultra_amd.graph import GraphedForward
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(dev)
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()
    triples = data.target_triples[:256]
    graphed = None
    with torch.no_grad():
        for s in range(0, 256, 8):
            for b in tasks.all_negative(data, triples[s:s + 8]):
                if graphed is None:
                    graphed = GraphedForward(model, data, b)
                got = graphed(b).clone()
                want = model(data, b)                         # eager forward between replays
                assert torch.equal(got, want), "replay %d diverged" % s
    a = ueval.evaluate(model, data, batch_size=8, max_triples=256, use_graph=True)
    b = ueval.evaluate(model, data, batch_size=8, max_triples=256, use_graph=False)
    assert a == b


def test_graph_survives_plan_cache_eviction_and_weight_updates(dev):
    """A captured forward holds raw pointers into its plans and into cached weight stacks: it keeps (and pins) the plans
    while the LRU plan cache turns over, and re-captures when a parameter is updated in place."""
    from tests.test_oracle_model import load_golden
    from ultra_amd import models, rspmm, synthetic, tasks
    from ultra_amd.graph import GraphedForward
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=2000, num_triple=16000, num_relation_base=9, num_test=64, seed=7).to(dev)
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()
    t_batch, _ = tasks.all_negative(data, data.target_triples[:8])
    with torch.no_grad():
        graphed = GraphedForward(model, data, t_batch)
        want = model(data, t_batch).clone()
        assert torch.equal(graphed(t_batch), want)
        # 40 other graphs push every plan of `data` out of the 16-entry cache (and their Plan objects are released)
        for seed in range(40):
            other = synthetic.make_kg(num_node=50, num_triple=200, num_relation_base=2, num_test=8, seed=100 + seed,
                                      relation_graph=False).to(dev)
            rspmm.get_plan(other.edge_index, other.edge_type, 50, 4).forward(
                torch.randn(4, 64, device=dev), torch.randn(50, 64, device=dev))
        torch.cuda.synchronize()
        assert all(p not in rspmm.cached_plans() for p in graphed._pinned)
        assert torch.equal(graphed(t_batch), want)
        # an optimizer-style in-place update: the replay must see the new weights (re-capture), not the cached stacks
        for prm in model.entity_model.layers[0].relation_projection.parameters():
            prm.mul_(1.05)
        model.entity_model.mlp[0].weight.add_(0.01)
        fresh = model(data, t_batch).clone()
        assert not torch.equal(fresh, want)
        assert torch.equal(graphed(t_batch), fresh)


def test_relation_graph_builder_matches_reference_golden(dev):
    """csrc/relgraph.hip against the relation graph the reference built (tests/golden/model_*.pt hold
    tasks.build_relation_graph's output for their KG): same edges, same order."""
    import os
    from ultra_amd import tasks
    from ultra_amd.data import Data
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "model_ultra_3g_sum.pt"))
    data = Data(edge_index=g["edge_index"], edge_type=g["edge_type"], num_nodes=g["num_nodes"], num_relations=g["num_relations"]).to(dev)
    tasks.build_relation_graph(data)
    assert torch.equal(data.relation_graph.edge_index.cpu(), g["rel_edge_index"])
    assert torch.equal(data.relation_graph.edge_type.cpu(), g["rel_edge_type"])


@pytest.mark.parametrize("shape", ["fb15k237", "wn18rr"])
def test_relation_graph_builder_at_dataset_shape_and_plan_format(dev, shape):
    """Edge for edge equal to the torch formulation on the CPU (itself pinned to the reference golden in tests/test_tasks.py);
    the device-built byte adjacency equals the one the host plan builder derives from the edge list; wall time printed
    beside the reference's 5.9 s (SURVEY.md section 8f-3, FB15k237 shape on 8 CPU cores)."""
    import time
    from ultra_amd import _lib, rspmm, synthetic, tasks
    cpu = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
    t0 = time.perf_counter()
    tasks.build_relation_graph(cpu)
    t_cpu = time.perf_counter() - t0
    gpu = cpu.to(dev)
    gpu.relation_graph = None
    tasks.build_relation_graph(gpu)          # warm-up (module load)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tasks.build_relation_graph(gpu)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    rg = gpu.relation_graph
    print("%s: relation graph %d nodes, %d edges; GPU %.2f ms, torch-on-CPU %.0f ms" %
          (shape, rg.num_nodes, rg.edge_index.shape[1], 1e3 * t_gpu, 1e3 * t_cpu))
    assert torch.equal(rg.edge_index.cpu(), cpu.relation_graph.edge_index)
    assert torch.equal(rg.edge_type.cpu(), cpu.relation_graph.edge_type)
    # plan format from the device
    dense = rspmm.Plan(cpu.relation_graph.edge_index, cpu.relation_graph.edge_type, rg.num_nodes, 4, exact_order=True).dense
    if dense is not None:
        want = dense.export(_lib.ARR_DENSE_ORDER)
        got = tasks.relation_graph_dense_adjacency(rg.adjacency_bits).cpu()
        assert torch.equal(got, want)


def test_captured_evaluation_step_equals_the_eager_protocol(dev):
    """evaluate() replays one hipGraph per full batch (graph.GraphedEvalStep: candidates, tail and head forward, both rank
    kernels; known answers listed once per shard) and runs the ragged last batch eagerly: same rankings, same metrics as
    the all-eager run, and the same again when evaluated a second time (fresh capture, plans re-pinned)."""
    from tests.test_oracle_model import load_golden
    from ultra_amd import eval as ueval
    from ultra_amd import models, synthetic
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=500, num_triple=4000, num_relation_base=5, num_test=45, seed=11).to(dev)
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()
    names = ("mr", "mrr", "hits@1", "hits@3", "hits@10", "hits@10_50", "mrr-tail")
    eager = ueval.evaluate(model, data, batch_size=8, metrics=names, use_graph=False)
    graphed = ueval.evaluate(model, data, batch_size=8, metrics=names)          # 5 replays + 1 eager batch of 5
    again = ueval.evaluate(model, data, batch_size=8, metrics=names)
    assert eager["_num_rankings"] == 90
    assert graphed == eager and again == eager
    # two captured steps in flight on two streams (the default from 128 batches on): same rankings, same metrics
    piped = ueval.evaluate(model, data, batch_size=8, metrics=names, in_flight=2)
    assert piped == eager
    # ... and three (evaluate()'s default from 256 batches on; bench.py's secondary.evaluate), with the time table filled in
    stats = {}
    piped = ueval.evaluate(model, data, batch_size=8, metrics=names, in_flight=3, stats=stats)
    assert piped == eager
    assert stats["in_flight"] == 3 and stats["batches"] == 5 and stats["capture"] > 0 and stats["replay"] > 0


def test_slot_stream_trial_of_evaluate_runs_on_real_batches_once_per_process(dev):
    """With enough batches evaluate() picks the streams of its captured steps by timing REAL batches (their rows are kept: the
    rankings equal the one-at-a-time protocol's) and keeps the choice for later calls of the process."""
    from tests.test_oracle_model import load_golden
    from ultra_amd import eval as ueval
    from ultra_amd import models, synthetic
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=300, num_triple=2500, num_relation_base=4, num_test=200, seed=12).to(dev)
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()
    eager = ueval.evaluate(model, data, batch_size=2, use_graph=False)
    ueval._SLOT_STREAMS.clear()
    stats = {}
    piped = ueval.evaluate(model, data, batch_size=2, in_flight=2, stats=stats)       # 100 batches: 4 candidates x 3 x 4 batches
    assert piped == eager
    assert stats["in_flight"] == 2 and 0 < stats["trial_batches"] <= 50 and len(stats["slot_streams"]["trial_ms"]) >= 2
    assert len(ueval._SLOT_STREAMS) == 1
    stats = {}
    again = ueval.evaluate(model, data, batch_size=2, in_flight=2, stats=stats)
    assert again == eager and "trial" not in stats                                    # the choice was kept
