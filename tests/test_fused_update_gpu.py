"""ultra_rspmm_forward_update: aggregate + layer update in one launch (the update runs in the tail of the reference-order
rspmm kernel, on the rows each workgroup has just summed) must equal the two launches it replaces -- ultra_rspmm_forward_point
then ultra_conv_update -- BIT FOR BIT, and with them the oracle chain (rspmm_oracle + the torch fp32 update)."""
import itertools

import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu

CASES = [
    dict(num_node=50, num_edge=400, num_relation=5, seed=0),
    dict(num_node=64, num_edge=300, num_relation=3, seed=1, hub=(7, 700)),          # a chain row
    dict(num_node=40, num_edge=100, num_relation=4, seed=2, empty_rows=10),         # rows without edges are updated too
    dict(num_node=1, num_edge=17, num_relation=2, seed=5),
    dict(num_node=300, num_edge=2000, num_relation=9, seed=8, hub=(11, 4321)),
    dict(num_node=3000, num_edge=40000, num_relation=40, seed=9, hub=(5, 3000)),    # several tiles per workgroup
]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _reset():
    from ultra_amd import rspmm
    rspmm.set_tuning()
    rspmm.set_plan_defaults()
    yield
    rspmm.set_tuning()
    rspmm.set_plan_defaults()


def _operands(case, batch, dev, seed):
    N, R = case["num_node"], case["num_relation"]
    g = torch.Generator().manual_seed(seed)
    rel = torch.randn(batch, R, 64, generator=g).to(dev)
    x = torch.randn(batch, N, 64, generator=g).to(dev)
    rows = torch.randint(0, N, (batch,), generator=g).to(dev)
    vals = torch.randn(batch, 64, generator=g).to(dev)
    weight = (torch.randn(64, 128, generator=g) / 11).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    ln_w, ln_b = torch.randn(64, generator=g).to(dev), torch.randn(64, generator=g).to(dev)
    return rel, x, rows, vals, weight, bias, ln_w, ln_b


FORMS = {"library": 0, "tail": 1, "lds": 3}     # rspmm.set_tuning(update_form=...): the library's choice (3 from 10 steps a row up, else 1) / in the kernel's tail / beside the walk, rows through LDS  (2 -- rows by reference -- was removed in round 5)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mul", ["mul", "add"])
@pytest.mark.parametrize("form", ["library", "tail", "lds"])
@pytest.mark.parametrize("batch,flags,point", [(8, 7, True), (3, 7, False), (1, 3, True), (2, 4, True), (5, 0, True), (16, 6, True)])
def test_one_launch_equals_the_two_launches(dev, case, mul, batch, flags, point, form):
    from ultra_amd import dense, rspmm
    from ultra_amd.rspmm import Plan
    ei, et = helpers.random_graph(**case)
    plan = Plan(ei, et, case["num_node"], case["num_relation"], exact_order=True)
    rel, x, rows, vals, weight, bias, ln_w, ln_b = _operands(case, batch, dev, case["seed"] + 100)
    pt = (rows, vals) if point else None
    agg = plan.forward(rel, x, sum="add", mul=mul, point=pt)
    want = dense._conv_update_forward(x, agg, weight, bias, ln_w if flags & 1 else None, ln_b if flags & 1 else None, 1e-5, flags)
    rspmm.set_tuning(update_form=FORMS[form])
    got = plan.forward_update(rel, x, weight, bias, ln_w if flags & 1 else None, ln_b if flags & 1 else None, 1e-5, flags,
                              mul=mul, point=pt)
    assert got is not None, "the stream walk serves this call"
    assert torch.equal(got, want), "max |d| = %g at %d elements" % ((got - want).abs().max().item(), int((got != want).sum()))
    again = plan.forward_update(rel, x, weight, bias, ln_w if flags & 1 else None, ln_b if flags & 1 else None, 1e-5, flags,
                                mul=mul, point=pt)
    assert torch.equal(again, want)


@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[4], CASES[5]])
@pytest.mark.parametrize("form", ["tail", "lds"])
@pytest.mark.parametrize("sum", ["max", "min"])
def test_max_aggregate_layer_in_one_launch(dev, case, form, sum):
    """BASELINE config 3 (max aggregate): the point boundary under max -- every other row meets 0, layers.py:206-207 -- and the
    update of the same launch, against the two launches on the boundary TENSOR."""
    from ultra_amd import dense, rspmm
    from ultra_amd.rspmm import Plan
    ei, et = helpers.random_graph(**case)
    N = case["num_node"]
    plan = Plan(ei, et, N, case["num_relation"], exact_order=True)
    batch = 4
    rel, x, rows, vals, weight, bias, ln_w, ln_b = _operands(case, batch, dev, case["seed"] + 7)
    bnd = torch.zeros(batch, N, 64, device=dev)
    bnd[torch.arange(batch, device=dev), rows] = vals
    agg = plan.forward(rel, x, sum=sum, mul="mul", boundary=bnd)
    want = dense._conv_update_forward(x, agg, weight, bias, ln_w, ln_b, 1e-5, 7)
    rspmm.set_tuning(update_form=FORMS[form])
    got = plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, mul="mul", point=(rows, vals), sum=sum)
    assert got is not None
    assert torch.equal(got, want), "max |d| = %g at %d elements" % ((got - want).abs().max().item(), int((got != want).sum()))


@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[3], CASES[5]])
@pytest.mark.parametrize("form", ["tail", "lds"])
@pytest.mark.parametrize("sum", ["add", "max"])
def test_one_launch_layer_against_the_oracle_chain(dev, case, form, sum):
    """Not through the two launches: the C oracle's rspmm (rspmm.cpp:50-75) on the host, the boundary as the reference
    combines it (layers.py:199-207), then torch's CPU Linear / LayerNorm / ReLU / residual (layers.py:233-240,
    models.py:158-160) -- bit for bit."""
    from oracle import rspmm_oracle
    from ultra_amd import rspmm
    from ultra_amd.rspmm import Plan
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    batch = 3
    rel, x, rows, vals, weight, bias, ln_w, ln_b = _operands(case, batch, dev, case["seed"] + 31)
    bnd = torch.zeros(batch, N, 64)
    bnd[torch.arange(batch), rows.cpu()] = vals.cpu()
    xn, reln, bndn = (t.cpu().transpose(0, 1).flatten(1).contiguous() for t in (x, rel, bnd))
    agg = rspmm_oracle.generalized_rspmm(ei, et, torch.ones(E), reln, xn, sum=sum, mul="mul")
    agg = agg + bndn if sum == "add" else torch.max(agg, bndn)
    agg = agg.view(N, batch, 64).transpose(0, 1)
    # nn.Linear / nn.LayerNorm in torch's CPU operation order as restated by oracle/torch_math_oracle.c -- pinned against
    # torch itself at the model's shapes (tests/test_torch_math.py); torch picks other GEMM kernels for the few rows of
    # these small graphs, which is why F.linear is not called here
    from oracle import torch_math_oracle as tm
    hidden = tm.linear(torch.cat([x.cpu(), agg], dim=-1).contiguous(), weight.cpu(), bias.cpu())
    want = torch.relu(tm.layer_norm(hidden, ln_w.cpu(), ln_b.cpu(), 1e-5)) + x.cpu()
    plan = Plan(ei, et, N, R, exact_order=True)
    rspmm.set_tuning(update_form=FORMS[form])
    got = plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, mul="mul", point=(rows, vals), sum=sum)
    assert got is not None
    assert torch.equal(got.cpu(), want), ((got.cpu() == want).float().mean().item(), (got.cpu() - want).abs().max().item())


def test_beside_the_walk_repeats_its_bits(dev):
    """200 launches of the update beside the walk give the same bits (the hand-off of rows between waves is not a race)."""
    from ultra_amd import dense, rspmm
    from ultra_amd.rspmm import Plan
    case = CASES[5]
    ei, et = helpers.random_graph(**case)
    plan = Plan(ei, et, case["num_node"], case["num_relation"], exact_order=True)
    rel, x, rows, vals, weight, bias, ln_w, ln_b = _operands(case, 8, dev, 77)
    agg = plan.forward(rel, x, sum="add", mul="mul", point=(rows, vals))
    want = dense._conv_update_forward(x, agg, weight, bias, ln_w, ln_b, 1e-5, 7)
    rspmm.set_tuning(update_form=3)
    assert plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals)) is not None
    outs = [plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals)) for _ in range(200)]
    torch.cuda.synchronize()
    assert all(torch.equal(o, want) for o in outs)


@pytest.mark.parametrize("grid", [8, 64, 256])
@pytest.mark.parametrize("form", ["tail", "lds"])
def test_any_number_of_workgroups_per_span(dev, grid, form):
    from ultra_amd import dense, rspmm
    from ultra_amd.rspmm import Plan
    case = CASES[4]
    ei, et = helpers.random_graph(**case)
    plan = Plan(ei, et, case["num_node"], case["num_relation"], exact_order=True)
    rel, x, rows, vals, weight, bias, ln_w, ln_b = _operands(case, 8, dev, 3)
    rspmm.set_tuning(grid=grid)
    agg = plan.forward(rel, x, sum="add", mul="mul", point=(rows, vals))
    want = dense._conv_update_forward(x, agg, weight, bias, ln_w, ln_b, 1e-5, 7)
    rspmm.set_tuning(grid=grid, update_form=FORMS[form])
    got = plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals))
    assert got is not None and torch.equal(got, want)


def test_calls_the_launch_does_not_serve_are_declined_not_approximated(dev):
    from ultra_amd.rspmm import Plan
    case = CASES[0]
    ei, et = helpers.random_graph(**case)
    rel, x, rows, vals, weight, bias, ln_w, ln_b = _operands(case, 2, dev, 4)
    loose = Plan(ei, et, case["num_node"], case["num_relation"], exact_order=False)
    assert loose.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals)) is None
    exact = Plan(ei, et, case["num_node"], case["num_relation"], exact_order=True)
    wide = torch.randn(2, case["num_node"], 128, device=dev)
    assert exact.forward_update(torch.randn(2, case["num_relation"], 128, device=dev), wide, weight, bias, ln_w, ln_b, 1e-5, 7) is None
    strided = torch.randn(2, case["num_node"], 128, device=dev)[:, :, :64]          # row stride 128: not the aggregate's stride
    assert exact.forward_update(rel, strided, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals)) is None
    from ultra_amd import rspmm
    rspmm.set_tuning(update_form=2)      # (round 4's by-reference form: removed in ABI 6 -- declined, never silently another form)
    assert exact.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals)) is None


@pytest.mark.parametrize("message_func", ["distmult", "transe"])
def test_layer_takes_the_one_launch_path_and_keeps_its_bits(dev, message_func):
    """GeneralizedRelationalConv (layers.py:84-131) on a point boundary, inference: FUSED_SPARSE_LAYER on / off."""
    from ultra_amd import layers
    case = CASES[4]
    ei, et = helpers.random_graph(**case)
    ei, et = ei.to(dev), et.to(dev)
    N, R = case["num_node"], case["num_relation"]
    torch.manual_seed(0)
    conv = layers.GeneralizedRelationalConv(64, 64, R, 64, message_func=message_func, aggregate_func="sum", layer_norm=True,
                                            activation="relu", dependent=False, project_relations=False).to(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, N, 64, generator=g).to(dev)
    query = torch.randn(4, 64, generator=g).to(dev)
    boundary = layers.PointBoundary(torch.tensor([3, 0, 77, 299], device=dev), torch.randn(4, 64, generator=g).to(dev), N)
    outs = []
    with torch.no_grad():
        for on in (True, False):
            layers.FUSED_SPARSE_LAYER = on
            try:
                outs.append(conv._forward_impl(x, query, boundary, ei, et, (N, N), residual=True))
            finally:
                layers.FUSED_SPARSE_LAYER = True
    assert torch.equal(outs[0], outs[1])


def test_form_3_on_the_random_schedules_of_the_plan_tests(dev):
    """the hand-off of form 3 on every random graph of tests/test_plan.py's schedule test (one node .. 2,000,
    no edges .. 30,000, hub rows of 257 .. 9,000 edges), on 8 and on 192 workgroups -- not only on the six fused cases above: bit
    for bit the two launches, and no bounded wait gave up (rspmm.check_device_error)."""
    import random
    from ultra_amd import dense, rspmm
    from ultra_amd.rspmm import Plan
    rng = random.Random(0)
    served = 0
    for it in range(14):
        N = rng.choice([1, 2, 7, 50, 300, 2000])
        E = rng.choice([1, 10, 500, 5000, 30000]) if N > 1 else rng.choice([1, 17])
        R = rng.choice([1, 3, 9])
        ei, et = helpers.random_graph(num_node=N, num_edge=E, num_relation=R, seed=it)
        for _ in range(rng.choice([0, 1, 3])):
            node, cnt = rng.randrange(N), rng.choice([257, 300, 700, 3000, 9000])
            g = torch.Generator().manual_seed(it * 7 + cnt)
            ei = torch.cat([ei, torch.stack([torch.full((cnt,), node), torch.randint(0, N, (cnt,), generator=g)])], dim=1)
            et = torch.cat([et, torch.randint(0, R, (cnt,), generator=g)])
        plan = Plan(ei, et, N, R, exact_order=True)
        case = dict(num_node=N, num_relation=R)
        rel, x, rows, vals, weight, bias, ln_w, ln_b = _operands(case, 8, dev, 100 + it)
        for sum in ("add", "max"):
            rspmm.set_tuning()
            agg = plan.forward(rel, x, sum=sum, mul="mul", point=(rows, vals))
            want = dense._conv_update_forward(x, agg, weight, bias, ln_w, ln_b, 1e-5, 7)
            for grid in (8, 192):
                rspmm.set_tuning(grid=grid, update_form=3)
                got = plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals), sum=sum)
                if got is None:      # (a schedule with more chain rows per workgroup than the control block lists: declined)
                    continue
                served += 1
                assert torch.equal(got, want), "graph %d, %s, grid %d" % (it, sum, grid)
    torch.cuda.synchronize()
    rspmm.check_device_error()
    assert served >= 40


def test_a_lost_arrival_ends_the_launch_with_an_error_not_a_hang(dev):
    """The waits of form 3's hand-off are bounded (OrderParams::err): with one update wave that never arrives at a meeting
    (CONV_DBG_LOSE_ARRIVAL, a fault injected for this test) the other three would spin for ever -- the launch must END, and the
    next look at the device error word must name the wait.  A clean launch afterwards is clean."""
    import time
    from ultra_amd import dense, rspmm
    from ultra_amd._lib import UltraError
    from ultra_amd.rspmm import Plan
    case = CASES[5]
    ei, et = helpers.random_graph(**case)
    plan = Plan(ei, et, case["num_node"], case["num_relation"], exact_order=True)
    rel, x, rows, vals, weight, bias, ln_w, ln_b = _operands(case, 2, dev, 5)
    agg = plan.forward(rel, x, sum="add", mul="mul", point=(rows, vals))
    want = dense._conv_update_forward(x, agg, weight, bias, ln_w, ln_b, 1e-5, 7)
    rspmm.set_tuning(update_form=3, grid=8)
    assert torch.equal(plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals)), want)
    torch.cuda.synchronize()
    rspmm.check_device_error()                               # nothing so far
    t0 = time.perf_counter()
    broken = plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7 | 1024, point=(rows, vals))
    torch.cuda.synchronize()                                 # returns: the launch ended
    assert broken is not None and time.perf_counter() - t0 < 60.0
    with pytest.raises(UltraError, match="bounded wait"):
        rspmm.check_device_error()
    rspmm.check_device_error()                               # the word was cleared by the look
    assert torch.equal(plan.forward_update(rel, x, weight, bias, ln_w, ln_b, 1e-5, 7, point=(rows, vals)), want)
    torch.cuda.synchronize()
    rspmm.check_device_error()
