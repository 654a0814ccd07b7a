"""Model-level parity on the GPU: Ultra.forward through the HIP engine vs (a) golden scores recorded
from the reference and (b) the CPU oracle model on a larger seeded graph.  Tolerance: fp32 scores
within 1e-4 (BASELINE north_star), rankings / metrics identical."""
import os

import pytest
import torch

from oracle import ultra_oracle_model
from tests.test_oracle_model import MODELS, load_golden
from ultra_amd import models, synthetic, tasks

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def build(state, cfg, dev):
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    return model.to(dev).eval()


@pytest.mark.parametrize("ckpt,aggr", MODELS)
def test_scores_match_reference_golden(dev, ckpt, aggr):
    g, state, data, cfg = load_golden(ckpt, aggr)
    model = build(state, cfg, dev)
    gdata = data.to(dev)
    batch = g["batch"].to(dev)
    pos_h, pos_t, _ = batch.t()
    with torch.no_grad():
        t_pred = model(gdata, g["t_batch"].to(dev))
        h_pred = model(gdata, g["h_batch"].to(dev))
        neg_pred = model(gdata, g["neg_batch"].to(dev))
        t_mask, h_mask = tasks.strict_negative_mask(gdata, batch)
        t_rank = tasks.compute_ranking(t_pred, pos_t, t_mask)
        h_rank = tasks.compute_ranking(h_pred, pos_h, h_mask)
    for got, want in ((t_pred, g["t_pred"]), (h_pred, g["h_pred"]), (neg_pred, g["neg_pred"])):
        err = (got.cpu() - want).abs().max().item()
        assert err <= TOL, "max |gpu - reference| = %g" % err
    assert torch.equal(t_rank.cpu(), g["t_rank"]) and torch.equal(h_rank.cpu(), g["h_rank"])


@pytest.mark.parametrize("ckpt,aggr", MODELS)
def test_scores_and_rankings_match_oracle_on_larger_graph(dev, ckpt, aggr):
    _, state, _, cfg = load_golden(ckpt, aggr)
    data = synthetic.make_kg(num_node=3000, num_triple=24000, num_relation_base=20, num_test=64, seed=11)
    batch = data.target_triples[:8]
    t_batch, h_batch = tasks.all_negative(data, batch)
    fn = ultra_oracle_model.reference_rspmm_fn()
    want_t = ultra_oracle_model.ultra_forward(state, cfg, data, t_batch, rspmm_fn=fn)
    want_h = ultra_oracle_model.ultra_forward(state, cfg, data, h_batch, rspmm_fn=fn)
    model = build(state, cfg, dev)
    gdata = data.to(dev)
    with torch.no_grad():
        got_t = model(gdata, t_batch.to(dev)).cpu()
        got_h = model(gdata, h_batch.to(dev)).cpu()
    assert (got_t - want_t).abs().max().item() <= TOL
    assert (got_h - want_h).abs().max().item() <= TOL
    t_mask, h_mask = tasks.strict_negative_mask(data, batch)
    pos_h, pos_t, _ = batch.t()
    for got, want, pos, mask in ((got_t, want_t, pos_t, t_mask), (got_h, want_h, pos_h, h_mask)):
        r_got = tasks.compute_ranking(got, pos, mask)
        r_want = tasks.compute_ranking(want, pos, mask)
        assert torch.equal(r_got, r_want), "rank mismatches: %d" % (r_got != r_want).sum().item()


def test_mean_and_transe_paths_run_and_match_oracle(dev):
    _, state, _, _ = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=400, num_triple=3000, num_relation_base=5, num_test=16, seed=2)
    batch = data.target_triples[:3]
    t_batch, _ = tasks.all_negative(data, batch)
    for aggr, msg in (("mean", "distmult"), ("sum", "transe"), ("max", "transe")):
        cfg = synthetic.default_model_cfg(aggregate_func=aggr, message_func=msg)
        want = ultra_oracle_model.ultra_forward(state, cfg, data, t_batch)
        model = build(state, cfg, dev)
        with torch.no_grad():
            got = model(data.to(dev), t_batch.to(dev)).cpu()
        assert (got - want).abs().max().item() <= TOL, (aggr, msg)


def test_training_step_gradients_match_cpu_autograd(dev):
    """fwd + bwd through the HIP autograd path vs torch autograd over the oracle model with a
    differentiable index_add restatement of rspmm (CPU, fp32)."""
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=300, num_triple=2400, num_relation_base=5, num_test=16, seed=4)
    batch = data.target_triples[:4]
    torch.manual_seed(0)
    neg = tasks.negative_sampling(data, batch, 8, strict=True)
    if os.environ.get("ULTRA_GRAD_PARITY_SHAPE"):
        # one-off long run (profiles/r6_experiments.txt): a BASELINE-sized graph, one query with 256 strict negatives -- ~ 5 GB of
        # saved fp64 messages on the CPU side
        data = synthetic.make_kg(**synthetic.SHAPES[os.environ["ULTRA_GRAD_PARITY_SHAPE"]], seed=1234)
        neg = tasks.negative_sampling(data, data.target_triples[:1], 256, strict=True)

    def torch_rspmm(edge_index, edge_type, edge_weight, relation, input, sum="add", mul="mul"):
        msg = relation[edge_type] * input[edge_index[1]] if mul == "mul" else relation[edge_type] + input[edge_index[1]]
        return torch.zeros_like(input).index_add(0, edge_index[0], msg * edge_weight.unsqueeze(-1))

    def cpu_grads(dtype):
        sd = {k: v.clone().to(dtype).requires_grad_() for k, v in state.items()}
        with torch.enable_grad():
            rel = ultra_oracle_model.rel_nbfnet(sd, data.relation_graph, neg[:, 0, 2], cfg["rel_model_cfg"], torch_rspmm)
            out = ultra_oracle_model.entity_nbfnet(sd, data, rel, neg, cfg["entity_model_cfg"], torch_rspmm)
            target = torch.zeros_like(out)
            target[:, 0] = 1
            loss = torch.nn.functional.binary_cross_entropy_with_logits(out, target)
            loss.backward()
        return loss.item(), {k: v.grad.double() for k, v in sd.items()}, target.float()

    loss32, g32, target = cpu_grads(torch.float32)      # the reference's arithmetic
    loss64, g64, _ = cpu_grads(torch.float64)           # what both fp32 runs approximate

    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()     # eval: keep the edge set fixed (remove_easy_edges is training-only)
    got = model(data.to(dev), neg.to(dev))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(got, target.to(dev))
    loss.backward()
    assert abs(loss.item() - loss32) <= 1e-5
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        scale = max(g64[name].abs().max().item(), 1e-6)
        err_gpu = (p.grad.cpu().double() - g64[name]).abs().max().item()
        err_cpu = (g32[name] - g64[name]).abs().max().item()
        # fp32 backprop through 12 layers: the GPU may not be further from the fp64 gradient than a few
        # times the reference's own fp32 path
        assert err_gpu <= 4 * err_cpu + 1e-4 * scale + 1e-7, \
            "%s: |gpu - fp64| = %g, |cpu fp32 - fp64| = %g (scale %g)" % (name, err_gpu, err_cpu, scale)


def test_training_mode_removes_easy_edges(dev):
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=200, num_triple=1500, num_relation_base=4, num_test=16, seed=6)
    # use graph edges as the batch so that remove_easy_edges has something to remove
    batch = torch.stack([data.edge_index[0, :4], data.edge_index[1, :4], data.edge_type[:4]], dim=-1)
    torch.manual_seed(0)
    neg = tasks.negative_sampling(data, batch, 4, strict=True)
    model = build(state, cfg, dev).train()
    out_train = model(data.to(dev), neg.to(dev))
    model.eval()
    with torch.no_grad():
        out_eval = model(data.to(dev), neg.to(dev))
    assert out_train.shape == out_eval.shape == (4, 5)
    assert not torch.allclose(out_train.detach(), out_eval)      # the direct edges were dropped in training mode


def test_graph_capture_replays_the_same_scores(dev):
    """hipGraph capture of the inference forward (ultra_amd/graph.py): replay == eager, for new batches too."""
    from ultra_amd.graph import GraphedForward
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=500, num_triple=4000, num_relation_base=6, num_test=64, seed=8).to(dev)
    model = build(state, cfg, dev)
    batches = [tasks.all_negative(data, data.target_triples[i * 4:(i + 1) * 4])[0] for i in range(3)]
    graphed = GraphedForward(model, data, batches[0])
    for b in batches:
        with torch.no_grad():
            want = model(data, b).clone()
        got = graphed(b, check=True).clone()
        assert torch.equal(got, want)
    with pytest.raises(ValueError):
        graphed(batches[0][:2])
    # the deferred input check still fires: rows must share head and relation
    bad = batches[0].clone()
    bad[0, 5, 0] += 1
    with pytest.raises(AssertionError):
        with torch.no_grad():
            model(data, bad)
    with pytest.raises(AssertionError):
        graphed(bad, check=True)


def test_query_nbfnet_matches_reference_golden(dev):
    """QueryNBFNet (UltraQuery's entity reasoner, models.py:212-275) on the same kernels."""
    import os
    from tests.test_oracle_model import GOLDEN
    from ultra_amd.data import Data
    g = torch.load(os.path.join(GOLDEN, "query_nbfnet_ultra_3g.pt"))
    state = torch.load(os.path.join(GOLDEN, "ultra_3g_model.pt"))
    cfg = synthetic.default_model_cfg()
    qnet = models.QueryNBFNet(**{k: v for k, v in cfg["entity_model_cfg"].items() if k != "class"})
    rnet = models.RelNBFNet(**{k: v for k, v in cfg["rel_model_cfg"].items() if k != "class"})
    qnet.load_state_dict({k[len("entity_model."):]: v for k, v in state.items() if k.startswith("entity_model.")})
    rnet.load_state_dict({k[len("relation_model."):]: v for k, v in state.items() if k.startswith("relation_model.")})
    qnet, rnet = qnet.to(dev).eval(), rnet.to(dev).eval()
    data = Data(edge_index=g["edge_index"], edge_type=g["edge_type"], num_nodes=g["num_nodes"],
                num_relations=g["num_relations"],
                relation_graph=Data(edge_index=g["rel_edge_index"], edge_type=g["rel_edge_type"],
                                    num_nodes=g["num_relations"], num_relations=4)).to(dev)
    with torch.no_grad():
        rel_repr = rnet(data.relation_graph, query=g["query_rels"].to(dev))
        assert (rel_repr.cpu() - g["rel_repr"]).abs().max().item() <= TOL
        query = rel_repr[torch.arange(3, device=dev), g["query_rels"].to(dev)]
        score = qnet(data, g["node_features"].to(dev), rel_repr, query)
    assert score.shape == g["score"].shape
    assert (score.cpu() - g["score"]).abs().max().item() <= TOL


def test_training_edge_dropout_by_weight_equals_edge_removal(dev):
    """sum aggregation: masking the batch's edges with 0/1 weights on the cached plan == rebuilding the graph
    without them (base_nbfnet.py:54-77), forward and gradients."""
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=200, num_triple=1500, num_relation_base=4, num_test=16, seed=6).to(dev)
    batch = torch.stack([data.edge_index[0, :4], data.edge_index[1, :4], data.edge_type[:4]], dim=-1)
    torch.manual_seed(0)
    neg = tasks.negative_sampling(data, batch, 4, strict=True)
    model = build(state, cfg, dev).train()
    ent = model.entity_model
    out_masked = model(data, neg)
    out_masked.sum().backward()
    g_masked = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    h, t, r = neg.unbind(-1)
    removed = ent.remove_easy_edges(data, h, t, r)
    assert removed.num_edges < data.num_edges
    ent.eval()                       # eval mode on the edge-removed graph = the reference's training forward
    out_removed = model(removed, neg)
    out_removed.sum().backward()
    ent.train()
    assert (out_masked - out_removed).abs().max().item() <= 1e-5
    for n, p in model.named_parameters():
        scale = max(p.grad.abs().max().item(), 1e-6)
        assert (p.grad - g_masked[n]).abs().max().item() <= 1e-4 * scale + 1e-7, n


@pytest.mark.parametrize("ckpt", ["ultra_3g"])
def test_onehot_layer0_path_matches_dense_path(dev, ckpt):
    """Layer 0 through the row-sparse forward == layer 0 through the dense forward (scores to fp32 round-off)."""
    from ultra_amd import layers
    _, state, _, cfg = load_golden(ckpt, "sum")
    data = synthetic.make_kg(num_node=1500, num_triple=20000, num_relation_base=6, num_test=16, seed=14).to(dev)
    model = build(state, cfg, dev)
    t_batch, h_batch = tasks.all_negative(data, data.target_triples[:6])
    try:
        with torch.no_grad():
            layers.POINT_BOUNDARY_FAST_PATH = False      # materialised boundary: the row-sparse rspmm serves layer 0
            layers.ONEHOT_FAST_PATH = True
            a_t, a_h = model(data, t_batch).clone(), model(data, h_batch).clone()
            layers.ONEHOT_FAST_PATH = False
            b_t, b_h = model(data, t_batch).clone(), model(data, h_batch).clone()
    finally:
        layers.ONEHOT_FAST_PATH = True
        layers.POINT_BOUNDARY_FAST_PATH = True
    assert (a_t - b_t).abs().max().item() <= 2e-5 and (a_h - b_h).abs().max().item() <= 2e-5


@pytest.mark.parametrize("ckpt,aggr", [("ultra_3g", "sum"), ("ultra_50g", "max")])
def test_closed_form_boundary_path_matches_materialised_boundary(dev, ckpt, aggr):
    """PointBoundary (layer 0 on its special rows, boundary added to one row per sample) == the (batch, N, d) tensor path.
    With the max aggregate the closed form does not apply (max(agg, 0) touches every row): both settings then take the
    tensor path and must agree exactly."""
    from ultra_amd import layers
    _, state, _, cfg = load_golden(ckpt, aggr)
    data = synthetic.make_kg(num_node=1500, num_triple=20000, num_relation_base=6, num_test=16, seed=15).to(dev)
    model = build(state, cfg, dev)
    t_batch, h_batch = tasks.all_negative(data, data.target_triples[:6])
    try:
        with torch.no_grad():
            layers.POINT_BOUNDARY_FAST_PATH = True
            a_t, a_h = model(data, t_batch).clone(), model(data, h_batch).clone()
            layers.POINT_BOUNDARY_FAST_PATH = False
            layers.ONEHOT_FAST_PATH = False
            b_t, b_h = model(data, t_batch).clone(), model(data, h_batch).clone()
    finally:
        layers.ONEHOT_FAST_PATH = True
        layers.POINT_BOUNDARY_FAST_PATH = True
    tol = 2e-5 if aggr == "sum" else 2e-5
    assert (a_t - b_t).abs().max().item() <= tol and (a_h - b_h).abs().max().item() <= tol


@pytest.mark.parametrize("bs", [1, 2, 5])
def test_every_combination_of_fast_path_switches_scores_the_same(dev, bs):
    """The four inference fast paths are independent: all 16 on/off combinations give the same scores as the generic
    path.  (Regression: with the prologue and the closed-form boundary both off, the strided head / relation index
    views reached a kernel through contiguous temporaries that were freed inside the argument list.)"""
    import itertools
    from ultra_amd import layers, models
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=1500, num_triple=20000, num_relation_base=6, num_test=16, seed=16).to(dev)
    model = build(state, cfg, dev)
    t_batch, h_batch = tasks.all_negative(data, data.target_triples[3:3 + bs])
    outs = {}
    try:
        for combo in itertools.product([True, False], repeat=4):
            layers.ONEHOT_FAST_PATH, layers.POINT_BOUNDARY_FAST_PATH, layers.FUSED_DENSE_LAYER, models.PROLOGUE_FAST_PATH = combo
            with torch.no_grad():
                outs[combo] = torch.cat([model(data, t_batch), model(data, h_batch)]).clone()
    finally:
        layers.ONEHOT_FAST_PATH = layers.POINT_BOUNDARY_FAST_PATH = layers.FUSED_DENSE_LAYER = True
        models.PROLOGUE_FAST_PATH = True
    base = outs[(False, False, False, False)]
    for combo, out in outs.items():
        err = (out - base).abs().max().item()
        assert err <= 3e-5, "switches %s: max |d| = %g" % (combo, err)


@pytest.mark.parametrize("dim,layers_n,layer_norm,short_cut,message", [(32, 3, True, True, "distmult"), (64, 2, False, False, "distmult"),
                                                                     (48, 2, True, False, "transe"), (128, 2, True, True, "distmult")])
def test_shapes_outside_the_fused_kernels_match_the_oracle(dev, dim, layers_n, layer_norm, short_cut, message):
    """Hidden sizes other than the checkpoints' 64 (and LayerNorm / short-cut variants) take the generic rspmm kernels with
    the torch update / readout chain; randomly initialised weights, scores against the CPU oracle model."""
    torch.manual_seed(dim + layers_n)
    def one(cls):
        return {"class": cls, "input_dim": dim, "hidden_dims": [dim] * layers_n, "message_func": message,
                "aggregate_func": "sum", "short_cut": short_cut, "layer_norm": layer_norm}
    cfg = {"rel_model_cfg": one("RelNBFNet"), "entity_model_cfg": one("EntityNBFNet")}
    model = models.Ultra(rel_model_cfg=dict(cfg["rel_model_cfg"]), entity_model_cfg=dict(cfg["entity_model_cfg"])).eval()
    data_cpu = synthetic.make_kg(num_node=400, num_triple=5000, num_relation_base=5, num_test=8, seed=dim)
    t_batch, h_batch = tasks.all_negative(data_cpu, data_cpu.target_triples[:3])
    want_t = ultra_oracle_model.ultra_forward(model.state_dict(), cfg, data_cpu, t_batch)
    want_h = ultra_oracle_model.ultra_forward(model.state_dict(), cfg, data_cpu, h_batch)
    data = data_cpu.to(dev)
    model = model.to(dev)
    with torch.no_grad():
        got_t = model(data, t_batch.to(dev)).cpu()
        got_h = model(data, h_batch.to(dev)).cpu()
    scale = max(1.0, want_t.abs().max().item(), want_h.abs().max().item())
    assert (got_t - want_t).abs().max().item() <= 1e-4 * scale
    assert (got_h - want_h).abs().max().item() <= 1e-4 * scale


def test_mixed_tail_and_head_rows_in_one_batch(dev):
    """base_nbfnet.py:79-86 converts head-prediction rows per ROW: one batch may hold both kinds (training batches do).
    Scores of a mixed batch == scores of the same rows scored in pure batches, on the fast and on the generic path."""
    from ultra_amd import layers
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=900, num_triple=9000, num_relation_base=5, num_test=16, seed=21).to(dev)
    model = build(state, cfg, dev)
    t_batch, h_batch = tasks.all_negative(data, data.target_triples[:6])
    mixed = torch.cat([t_batch[:2], h_batch[2:4], t_batch[4:5], h_batch[5:6]])
    with torch.no_grad():
        want = torch.cat([model(data, t_batch)[:2], model(data, h_batch)[2:4], model(data, t_batch)[4:5], model(data, h_batch)[5:6]])
        got = model(data, mixed)
        try:
            models.PROLOGUE_FAST_PATH = False
            layers.POINT_BOUNDARY_FAST_PATH = False
            generic = model(data, mixed)
        finally:
            models.PROLOGUE_FAST_PATH = True
            layers.POINT_BOUNDARY_FAST_PATH = True
    assert (got - want).abs().max().item() <= 2e-5
    assert (generic - want).abs().max().item() <= 2e-5


def test_sparse_relation_graph_takes_the_edge_walk(dev):
    """Many relations, few triples: the relation graph is far from complete, gets no dense-format plan and runs through the
    edge-walk kernels (point boundary, layer-0 kernels); scores against the CPU oracle model."""
    from ultra_amd import rspmm
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data_cpu = synthetic.make_kg(num_node=500, num_triple=1500, num_relation_base=150, num_test=8, seed=23)
    rg = data_cpu.relation_graph
    fill = rg.num_edges / float(rg.num_nodes ** 2 * 4)
    assert fill < 0.25, fill
    t_batch, h_batch = tasks.all_negative(data_cpu, data_cpu.target_triples[:4])
    want_t = ultra_oracle_model.ultra_forward(state, cfg, data_cpu, t_batch)
    want_h = ultra_oracle_model.ultra_forward(state, cfg, data_cpu, h_batch)
    data = data_cpu.to(dev)
    model = build(state, cfg, dev)
    with torch.no_grad():
        got_t, got_h = model(data, t_batch.to(dev)).cpu(), model(data, h_batch.to(dev)).cpu()
    assert rspmm.get_plan(data.relation_graph.edge_index, data.relation_graph.edge_type, rg.num_nodes, 4).dense is None
    assert (got_t - want_t).abs().max().item() <= 1e-4 and (got_h - want_h).abs().max().item() <= 1e-4


@pytest.mark.parametrize("threshold,key", [(0.0, "t_prob"), (0.3, "t_prob_thr03")])
def test_relation_projection_matches_reference_golden(dev, threshold, key):
    """RelationProjection (ultraquery.py:245-277) against the output of the reference class itself (executed from the
    reference file by tests/golden/gen_golden.py around the reference's Ultra(RelNBFNet, QueryNBFNet))."""
    import os

    from ultra_amd.data import Data
    from ultra_amd.ultraquery import RelationProjection
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "relation_projection_ultra_3g.pt"))
    state = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ultra_3g_model.pt"))
    rel_graph = Data(edge_index=g["rel_edge_index"], edge_type=g["rel_edge_type"], num_nodes=g["num_relations"],
                     num_relations=4)
    data = Data(edge_index=g["edge_index"], edge_type=g["edge_type"], num_nodes=g["num_nodes"],
                num_relations=g["num_relations"], relation_graph=rel_graph).to(dev)
    cfg = synthetic.default_model_cfg()
    cfg["entity_model_cfg"]["class"] = "QueryNBFNet"
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    proj = RelationProjection(model.to(dev).eval(), threshold=threshold)
    with torch.no_grad():
        got = proj(data, g["h_prob"].to(dev), g["r_index"].to(dev)).cpu()
    assert got.shape == g[key].shape
    assert (got - g[key]).abs().max().item() <= 1e-5, (got - g[key]).abs().max().item()


@pytest.mark.parametrize("share_chip", ["auto", True, False])
def test_two_batches_in_flight_score_like_one_at_a_time(dev, share_chip):
    """graph.PipelinedForward: captured forwards replayed round-robin on two streams (own buffers each) return, batch for
    batch, the bits of the one-at-a-time forward -- with whole-chip aggregation launches and with the shared-chip ones (three
    quarters of the CUs as workgroups: the schedule for fewer partitions moves rows between workgroups, not inside their sums)."""
    from ultra_amd import graph as ugraph, models, rspmm, synthetic, tasks
    data = synthetic.make_kg(num_node=900, num_triple=6000, num_relation_base=11, seed=5).to(dev)
    torch.manual_seed(3)
    model = models.Ultra(**synthetic.default_model_cfg()).to(dev).eval()
    bs = 4
    batches = [tasks.all_negative(data, data.target_triples[i * bs:(i + 1) * bs])[0] for i in range(6)]
    with torch.no_grad():
        want = [model(data, b).clone() for b in batches]
    piped = ugraph.PipelinedForward(model, data, batches[0], depth=2, share_chip=share_chip)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    assert piped.launch_grid == (0 if share_chip is False else ugraph.shared_launch_grid(dev))      # (a small graph: "auto" shares)
    assert 0 < ugraph.shared_launch_grid(dev) < cus
    t = rspmm.get_tuning()
    assert t["grid"] == 0, "the launch size of the captures must not leak into the process's tuning"
    for rep in range(3):
        got = []
        for i, b in enumerate(batches):
            out = piped(b)
            if i >= 1:          # the previous slot's output: still in place (the slot is reused two calls later)
                piped.join()
                got.append((i - 1, prev.clone()))
            prev = out
        piped.join()
        got.append((len(batches) - 1, prev.clone()))
        for i, g in got:
            assert torch.equal(g, want[i]), "batch %d, repetition %d" % (i, rep)
    # back to back without joining in between: only the last two are still readable
    for b in batches:
        out = piped(b)
    piped.join()
    assert torch.equal(out, want[-1])
