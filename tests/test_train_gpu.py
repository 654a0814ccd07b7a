"""ultra_amd/train.py on the GPU: the one-launch loss against the reference's op chain (script/run.py:66-77) and the captured
training step against the same steps launched one by one."""
import pytest
import torch
from torch.nn import functional as F

from tests.test_oracle_model import load_golden
from ultra_amd import layers, models, synthetic, tasks, train

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def reference_loss(pred, temperature, num_negative):
    """script/run.py:66-77, verbatim semantics."""
    target = torch.zeros_like(pred)
    target[:, 0] = 1
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction="none")
    neg_weight = torch.ones_like(pred)
    if temperature > 0:
        with torch.no_grad():
            neg_weight[:, 1:] = F.softmax(pred[:, 1:] / temperature, dim=-1)
    else:
        neg_weight[:, 1:] = 1 / num_negative
    loss = (loss * neg_weight).sum(dim=-1) / neg_weight.sum(dim=-1)
    return loss.mean()


@pytest.mark.parametrize("rows,n", [(1, 2), (8, 257), (3, 65), (64, 33), (16, 1025)])
@pytest.mark.parametrize("temperature", [1.0, 0.5, 0.0])
def test_ranking_loss_kernel_matches_the_reference_op_chain(dev, rows, n, temperature):
    gen = torch.Generator().manual_seed(rows * 1000 + n)
    pred = (torch.randn(rows, n, generator=gen) * 4).to(dev)
    pred[0, 0] = 30.0
    pred[-1, -1] = -40.0          # saturated logits: log_sigmoid must not overflow
    want_in = pred.double().clone().requires_grad_()
    want = reference_loss(want_in, temperature, n - 1)
    want.backward()
    got_in = pred.clone().requires_grad_()
    got = train.ranking_loss(got_in, temperature, n - 1)
    assert type(got.grad_fn).__name__ == "_RankingLossBackward"
    (got * 3).backward()           # (the node scales its stored gradient by the incoming one)
    assert abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    scale = want_in.grad.abs().max().item()
    assert (got_in.grad.double() / 3 - want_in.grad).abs().max().item() <= 2e-6 * scale + 1e-9
    # the torch route of the same function
    train.FUSED_LOSS = False
    try:
        plain = train.ranking_loss(pred, temperature, n - 1)
    finally:
        train.FUSED_LOSS = True
    assert abs(plain.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))


def _setup(dev, seed=4):
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=700, num_triple=6000, num_relation_base=9, num_test=16, seed=seed).to(dev)
    triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)[: data.num_edges // 2]
    torch.manual_seed(seed)
    batches = [tasks.negative_sampling(data, triples[8 * i:8 * i + 8], 32, strict=True) for i in range(6)]

    def fresh():
        model = models.Ultra(**cfg)
        model.load_state_dict(state)
        return model.to(dev).train()
    return data, batches, fresh


@pytest.mark.parametrize("rows_route", [False, True])
def test_captured_training_step_equals_the_steps_launched_one_by_one(dev, rows_route):
    """Six AdamW steps through GraphedTrainStep against train_step on the same batches from the same start: the same losses and
    the same parameters afterwards, bit for bit (every sum of the step has a fixed order, with the last layer on the candidates'
    rows or as a whole-graph walk) -- with eager work on another stream between the replays."""
    was = layers.LAST_LAYER_ON_ROWS
    layers.LAST_LAYER_ON_ROWS = rows_route
    try:
        data, batches, fresh = _setup(dev)
        eager = fresh()
        opt = train.make_adamw(eager, lr=5e-3)
        want_loss = [train.train_step(eager, data, b, opt, num_negative=32).item() for b in batches]

        model = fresh()
        before = [p.detach().clone() for p in model.parameters()]
        opt = train.make_adamw(model, lr=5e-3, capturable=True)
        step = train.GraphedTrainStep(model, data, opt, batches[0], num_negative=32)
        for p, b in zip(model.parameters(), before):
            assert torch.equal(p, b)            # the warm-up steps left no trace
        side = torch.cuda.Stream()
        got_loss = []
        for b in batches:
            with torch.cuda.stream(side):       # unrelated eager work beside the replays (allocations, fills, reductions)
                junk = torch.zeros(1 << 20, device=dev).add_(1).sum()
            got_loss.append(step(b).item())
            step.check()
        del junk
        # bit for bit on both routes: since round 6 the last layer's backward on the candidates' rows is a gather in a fixed order
        assert got_loss == want_loss, (got_loss, want_loss)
        for (name, p), q in zip(model.named_parameters(), eager.parameters()):
            assert torch.equal(p, q), name
        assert int(next(iter(opt.state.values()))["step"].item()) == len(batches)
    finally:
        layers.LAST_LAYER_ON_ROWS = was


def test_captured_step_refuses_another_batch_shape_and_an_uncapturable_optimizer(dev):
    data, batches, fresh = _setup(dev)
    model = fresh()
    with pytest.raises(ValueError):
        train.GraphedTrainStep(model, data, train.make_adamw(model), batches[0])
    step = train.GraphedTrainStep(model, data, train.make_adamw(model, capturable=True), batches[0], num_negative=32)
    with pytest.raises(ValueError):
        step(batches[0][:4])


@pytest.mark.parametrize("num_node,expect_rows", [(4 * 33, True), (4 * 33 - 1, False)])
def test_rows_route_threshold_is_a_quarter_of_the_graph(dev, num_node, expect_rows):
    """layers.training_rows_layer takes the last layer on the candidates' rows iff 4 * (1 + num_negative) <= num_node (a list that
    covers most of the graph takes the whole layer's walk): pinned on both sides of the threshold."""
    _, state, _, cfg = load_golden("ultra_3g", "sum")
    data = synthetic.make_kg(num_node=num_node, num_triple=1500, num_relation_base=4, num_test=16, seed=3).to(dev)
    batch = torch.stack([data.edge_index[0, :4], data.edge_index[1, :4], data.edge_type[:4]], dim=-1)
    torch.manual_seed(0)
    neg = tasks.negative_sampling(data, batch, 32, strict=False)          # 33 candidates per row
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).train()
    out = model(data, neg)
    assert model.entity_model._last_hidden_on_rows == expect_rows
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
