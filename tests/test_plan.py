"""Host logic of the plan builder (no GPU): structure invariants + emulated walk vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import rspmm_oracle
from tests import helpers
from ultra_amd import _lib
from ultra_amd.rspmm import Plan

CASES = [
    dict(num_node=50, num_edge=400, num_relation=5, seed=0),
    dict(num_node=64, num_edge=300, num_relation=3, seed=1, hub=(7, 700)),           # hub row -> split + wave items
    dict(num_node=40, num_edge=100, num_relation=4, seed=2, empty_rows=10),          # empty rows keep the identity
    dict(num_node=30, num_edge=200, num_relation=1, seed=3, duplicates=50),          # single relation, duplicate edges
    dict(num_node=5, num_edge=0, num_relation=2, seed=4),                             # no edges at all
    dict(num_node=1, num_edge=17, num_relation=2, seed=5),                            # one node, self loops
]


def _plan_arrays(plan):
    return {k: plan.export(v).numpy() for k, v in
            dict(row_ptr=_lib.ARR_ROW_PTR, col=_lib.ARR_COL, type=_lib.ARR_TYPE, perm=_lib.ARR_PERM,
                 item=_lib.ARR_ITEM, split_row=_lib.ARR_SPLIT_ROW, split_ptr=_lib.ARR_SPLIT_PTR).items()}


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("opts", [dict(), dict(seg_len=16, g_max=4), dict(seg_len=64, g_max=64), dict(exact_order=True)])
def test_plan_structure(case, opts):
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    plan = Plan(ei, et, N, R, **opts)
    info = plan.info()
    a = _plan_arrays(plan)
    assert info["num_edge"] == E and info["num_node"] == N
    # CSR: sorted by (row, col), stable in the original edge id; perm is a permutation
    perm = a["perm"]
    assert sorted(perm.tolist()) == list(range(E))
    rows = ei[0].numpy()[perm]
    cols = ei[1].numpy()[perm]
    key = rows.astype(np.int64) * (N + 1) + cols
    assert (np.diff(key) >= 0).all()
    same = np.diff(key) == 0
    assert (np.diff(perm)[same] > 0).all(), "ties must keep the original order"
    assert (a["col"] == cols).all() and (a["type"] == et.numpy()[perm]).all()
    assert a["row_ptr"][0] == 0 and a["row_ptr"][-1] == E
    assert (np.bincount(rows, minlength=N) == np.diff(a["row_ptr"])).all()
    # items: cover every edge exactly once, stay inside their row, honour seg_len / g_max classes
    items = a["item"].reshape(-1, 4)
    covered = np.zeros(E, dtype=np.int64)
    for idx, (row, begin, length, slot) in enumerate(items):
        assert a["row_ptr"][row] <= begin and begin + length <= a["row_ptr"][row + 1]
        covered[begin:begin + length] += 1
        if not opts.get("exact_order"):
            assert length <= info["seg_len"]
            assert (length > info["g_max"]) == (idx < info["n_wave_item"])
    assert (covered == 1).all()
    assert set(items[:, 0].tolist()) == set(range(N)), "rows without edges still own an (empty) item"
    nw = info["n_wave_item"]
    assert (np.diff(items[:nw, 2]) <= 0).all() and (np.diff(items[nw:, 2]) <= 0).all(), "descending length per class"
    # split rows: slots are contiguous per row and ordered by edge position
    slots = items[items[:, 3] >= 0]
    assert len(slots) == info["n_partial_slot"]
    for k, row in enumerate(a["split_row"]):
        mine = slots[slots[:, 0] == row]
        mine = mine[np.argsort(mine[:, 3])]
        assert mine[:, 3].tolist() == list(range(a["split_ptr"][k], a["split_ptr"][k + 1]))
        assert (np.diff(mine[:, 1]) > 0).all()
        assert mine[:, 2].sum() == a["row_ptr"][row + 1] - a["row_ptr"][row]
    if opts.get("exact_order"):
        assert info["n_wave_item"] == 0 and info["n_partial_slot"] == 0


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("sum", ["add", "min", "max"])
@pytest.mark.parametrize("mul", ["mul", "add"])
def test_emulated_walk_matches_oracle(case, sum, mul):
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    rel, x, w = helpers.features(N, R, 24, E, seed=case["seed"])
    want = rspmm_oracle.generalized_rspmm(ei, et, w, rel, x, sum=sum, mul=mul)
    for opts in (dict(seg_len=16, g_max=4), dict()):
        plan = Plan(ei, et, N, R, **opts)
        got = helpers.emulate_plan_forward(plan, rel, x, edge_weight=w, sum=sum, mul=mul)
        if sum == "add":
            helpers.assert_sum_close(got, want, ei, et, w, rel, x, mul=mul)
        else:
            assert torch.equal(got, want), "min/max are order independent: must be bit exact"
    # exact-order plan: sequential walk in (row, col) order == the oracle's order, bit for bit
    plan = Plan(ei, et, N, R, exact_order=True)
    got = helpers.emulate_plan_forward(plan, rel, x, edge_weight=w, sum=sum, mul=mul)
    assert torch.equal(got, want)


def test_plan_rejects_bad_input():
    ei, et = helpers.random_graph(10, 20, 3)
    with pytest.raises(RuntimeError):
        Plan(ei, et, 5, 3)          # node id out of range
    with pytest.raises(RuntimeError):
        Plan(ei, et, 10, 2)         # relation id out of range
    with pytest.raises(RuntimeError):
        Plan(ei[0], et, 10, 3)      # edge_index must be (2, E)
    with pytest.raises(RuntimeError):
        Plan(ei, et[:-1], 10, 3)
    with pytest.raises(RuntimeError):
        Plan(ei, et.int(), 10, 3)   # checkSameType(edge_index, edge_type)


DENSE = dict(num_node=40, num_edge=3000, num_relation=4, seed=9)      # relation-graph regime: long (row, type) runs


@pytest.mark.parametrize("case", [DENSE, CASES[1], CASES[3], CASES[4]])
def test_type_run_plan_structure(case):
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    plan = Plan(ei, et, N, R, seg_len=32, g_max=8, type_runs="only")
    info = plan.info()
    a = _plan_arrays(plan)
    assert info["flags"] & _lib.PLAN_TYPE_RUNS
    perm = a["perm"]
    rows, cols, types = ei[0].numpy()[perm], ei[1].numpy()[perm], et.numpy()[perm]
    key = (rows.astype(np.int64) * R + types) * (N + 1) + cols
    assert (np.diff(key) >= 0).all(), "sorted by (row, type, col)"
    items = a["item"].reshape(-1, 4)
    covered = np.zeros(E, dtype=np.int64)
    per_row = np.zeros(N, dtype=np.int64)
    for row, begin, length, slot in items:
        covered[begin:begin + length] += 1
        per_row[row] += 1
        assert length <= 32
        if length:
            assert len(set(types[begin:begin + length].tolist())) == 1, "one relation per item"
            assert (rows[begin:begin + length] == row).all()
    assert (covered == 1).all() and (per_row >= 1).all()
    for row, begin, length, slot in items:
        assert (slot >= 0) == (per_row[row] > 1), "rows with several items combine through partial slots"
    runs = len(set(zip(ei[0].tolist(), et.tolist())))
    assert info["n_type_run"] == runs
    assert Plan(ei, et, N, R).info()["n_type_run"] == runs           # the statistic does not depend on the sort order


def test_type_run_twin_is_automatic_for_dense_few_relation_graphs():
    ei, et = helpers.random_graph(**DENSE)
    assert Plan(ei, et, 40, 4).typed is not None                      # mean run length 3000 / 160 ~ 19
    ei2, et2 = helpers.random_graph(**CASES[0])
    assert Plan(ei2, et2, 50, 5).typed is None                        # 400 edges over ~200 runs
    assert Plan(ei, et, 40, 4, type_runs=False).typed is None
    assert Plan(ei, et, 40, 4, exact_order=True).typed is None


@pytest.mark.parametrize("case", [DENSE, CASES[3]])
def test_type_run_emulated_walk_matches_oracle(case):
    ei, et = helpers.random_graph(**case)
    N, R, E = case["num_node"], case["num_relation"], ei.shape[1]
    rel, x, w = helpers.features(N, R, 24, E, seed=3)
    plan = Plan(ei, et, N, R, seg_len=32, g_max=8, type_runs=True)
    assert plan.typed is not None
    for weight in (w, None):
        want = rspmm_oracle.generalized_rspmm(ei, et, w if weight is not None else torch.ones(E), rel, x)
        got = helpers.emulate_plan_forward(plan, rel, x, edge_weight=weight, sum="add", mul="mul")
        helpers.assert_sum_close(got, want, ei, et, w if weight is not None else torch.ones(E), rel, x)
    # other semirings keep using the (row, col) plan
    want = rspmm_oracle.generalized_rspmm(ei, et, w, rel, x, sum="max", mul="mul")
    assert torch.equal(helpers.emulate_plan_forward(plan, rel, x, edge_weight=w, sum="max", mul="mul"), want)


def _dense_from_fragments(plan, N, R):
    """Undo the MFMA A-operand order of ULTRA_ARR_DENSE: -> (R, N, N) multiplicities."""
    frag = plan.export(_lib.ARR_DENSE)
    n_rt, kg = (N + 31) // 32, ((N + 7) // 8 + 19) // 20 * 20          # k groups padded to whole pipeline blocks
    tc = 1 if R <= 1 else (2 if R == 2 else 4)
    ntc = (R + tc - 1) // tc
    assert frag.dtype == torch.uint8 and frag.numel() == n_rt * ntc * kg * 64 * tc * 4 == plan.info()["dense_bytes"]
    frag = frag.view(n_rt, ntc, kg, 64, tc, 4).float()         # [row tile][type chunk][k group][lane][type in chunk][q]
    lane = torch.arange(64)
    i, h = lane % 32, lane // 32
    dense = torch.zeros(ntc * tc, n_rt * 32, kg * 8)
    for rt in range(n_rt):
        for c in range(ntc):
            for g in range(kg):
                for tl in range(tc):
                    for q in range(4):
                        # row = 32 row_tile + lane % 32, column = 8 kgroup + 2 q + lane // 32
                        dense[c * tc + tl, rt * 32 + i, g * 8 + 2 * q + h] = frag[rt, c, g, :, tl, q]
    assert dense[R:].abs().sum() == 0 and dense[:, N:].abs().sum() == 0 and dense[:, :, N:].abs().sum() == 0, \
        "padding cells must stay zero"
    return dense[:R, :N, :N]


@pytest.mark.parametrize("case", [DENSE, dict(num_node=70, num_edge=6000, num_relation=2, seed=5, duplicates=300),
                                  dict(num_node=33, num_edge=2000, num_relation=1, seed=6),
                                  dict(num_node=20, num_edge=3000, num_relation=7, seed=7)])
def test_dense_format_plan_holds_the_edge_multiplicities(case):
    ei, et = helpers.random_graph(**case)
    N, R = case["num_node"], case["num_relation"]
    plan = Plan(ei, et, N, R, dense=True)
    assert plan.dense is not None
    got = _dense_from_fragments(plan.dense, N, R)
    want = torch.zeros(R, N, N)
    want.index_put_((et, ei[0], ei[1]), torch.ones(ei.shape[1]), accumulate=True)
    assert torch.equal(got, want)


def test_dense_format_needs_multiplicities_that_fit_a_byte():
    ei = torch.zeros(2, 300, dtype=torch.long)                        # one edge repeated 300 times
    et = torch.zeros(300, dtype=torch.long)
    assert Plan(ei, et, 4, 1).dense is None                           # "auto": the edge walk serves it
    with pytest.raises(RuntimeError):
        Plan(ei, et, 4, 1, dense=True)
    ok = Plan(ei[:, :255], et[:255], 4, 1, dense=True)
    assert _dense_from_fragments(ok.dense, 4, 1)[0, 0, 0] == 255


def test_dense_twin_is_automatic_only_for_filled_graphs():
    ei, et = helpers.random_graph(**DENSE)                            # 3000 edges over 40 * 40 * 4 cells
    assert Plan(ei, et, 40, 4).dense is not None
    ei2, et2 = helpers.random_graph(**CASES[0])                       # 400 edges over 50 * 50 * 5 cells
    assert Plan(ei2, et2, 50, 5).dense is None
    assert Plan(ei, et, 40, 4, dense=False).dense is None
    assert Plan(ei, et, 40, 4, exact_order=True).dense is None
    with pytest.raises(RuntimeError):                                 # beyond ULTRA_DENSE_MAX_IN_ROW
        big = torch.zeros(2, 1, dtype=torch.long)
        Plan(big, torch.zeros(1, dtype=torch.long), 2000, 1, dense="only")


def test_group_streams_cover_every_group_row_once_in_sorted_edge_order():
    """Schedule of the assembly walk (plan.cpp build_schedule): every non-chain row sits in exactly one stream, its
    records are the row's sorted edges followed by a marker (row, num_relation), descriptors tile the record array."""
    from ultra_amd.rspmm import Plan
    from ultra_amd import _lib
    ei, et = helpers.random_graph(num_node=300, num_edge=2000, num_relation=9, seed=8, hub=(11, 700))
    N, R = 300, 9
    plan = Plan(ei, et, N, R, exact_order=True)
    nparts = 4
    sdesc, srec = plan.streams(nparts)
    assert sdesc.shape == (nparts * 64, 2)
    row_ptr = plan.export(_lib.ARR_ROW_PTR)
    col, typ = plan.export(_lib.ARR_COL), plan.export(_lib.ARR_TYPE)
    n_chain = plan.info()["n_chain_row"]
    items = plan.export(_lib.ARR_ITEM).view(-1, 4)
    chain_rows = set(items[:n_chain, 0].tolist())
    seen = set()
    pos = 0
    for g in range(nparts * 64):
        begin, steps = sdesc[g].tolist()
        assert begin == pos
        k = begin
        while k < begin + steps:
            # next marker
            m = k
            while srec[m, 1] != R:
                m += 1
            row = int(srec[m, 0])
            assert row not in seen and row not in chain_rows
            seen.add(row)
            b, e = int(row_ptr[row]), int(row_ptr[row + 1])
            assert m - k == e - b
            assert torch.equal(srec[k:m, 0], col[b:e]) and torch.equal(srec[k:m, 1], typ[b:e])
            k = m + 1
        assert k == begin + steps
        pos = begin + steps
    assert pos == srec.shape[0]
    assert seen == set(range(N)) - chain_rows
    loads = sdesc[:, 1].view(nparts, 64)
    assert int(loads.max() - loads.min()) <= int((row_ptr[1:] - row_ptr[:-1]).clamp(max=256).max()) + 64


def test_twelve_walker_schedule_leaves_the_update_waves_without_rows():
    """The schedule of the launches whose last four waves apply the layer update beside the walk (rspmm_order_kernel,
    UPDATE == 2): streams 48..63 of every workgroup are empty, every non-chain row still sits in exactly one stream."""
    from ultra_amd.rspmm import Plan
    from ultra_amd import _lib
    ei, et = helpers.random_graph(num_node=3000, num_edge=40000, num_relation=9, seed=8, hub=(11, 700))
    N, R = 3000, 9
    plan = Plan(ei, et, N, R, exact_order=True)
    nparts = 4
    sdesc, srec = plan.streams(nparts, walkers=12)
    per = sdesc[:, 1].view(nparts, 64)
    assert int(per[:, 48:].sum()) == 0 and int(per[:, :48].min()) > 0
    n_chain = plan.info()["n_chain_row"]
    chain_rows = set(plan.export(_lib.ARR_ITEM).view(-1, 4)[:n_chain, 0].tolist())
    markers = srec[srec[:, 1] == R, 0].tolist()
    assert len(markers) == len(set(markers)) and set(markers) == set(range(N)) - chain_rows
    # the sixteen-walker schedule of the same plan is untouched by the request
    sdesc16, _ = plan.streams(nparts)
    assert int(sdesc16[:, 1].view(nparts, 64)[:, 48:].min()) > 0


def _schedule_array(plan, key, which):
    import ctypes
    from ultra_amd import _lib
    n = ctypes.c_int64()
    _lib.check(_lib.lib.ultra_plan_schedule_export(plan._h, key, which, None, 0, ctypes.byref(n)))
    t = torch.empty(n.value, dtype=torch.int32)
    _lib.check(_lib.lib.ultra_plan_schedule_export(plan._h, key, which, t.data_ptr(), n.value, ctypes.byref(n)))
    return t


def test_twelve_walker_schedule_walks_shorter_chain_rows_as_stream_rows_longest_last():
    """plan.cpp CHAIN_LIMIT_FACTOR: in the schedules of the update-beside-the-walk launches a row the plan lists as a chain row
    (> 256 edges) is a stream row if it is at most 2.1 x the mean stream length; longer rows stay chains; every row is still
    served exactly once; inside a stream the rows come shortest first (the longest last).  The sixteen-walker schedule of the
    same plan keeps every listed chain row a chain row (it also serves the unit walk)."""
    from ultra_amd.rspmm import Plan
    from ultra_amd import _lib
    N, R = 3000, 9
    ei, et = helpers.random_graph(num_node=N, num_edge=40000, num_relation=R, seed=8, hub=(11, 700))
    extra_rows = torch.cat([torch.full((300,), 5), torch.full((6000,), 7)])            # a 300-odd and a 6,000-odd edge row too
    g = torch.Generator().manual_seed(1)
    ei = torch.cat([ei, torch.stack([extra_rows, torch.randint(0, N, (6300,), generator=g)])], dim=1)
    et = torch.cat([et, torch.randint(0, R, (6300,), generator=g)])
    plan = Plan(ei, et, N, R, exact_order=True)
    n_chain = plan.info()["n_chain_row"]
    items = plan.export(_lib.ARR_ITEM).view(-1, 4)
    listed = {int(r): int(l) for r, l in zip(items[:n_chain, 0].tolist(), items[:n_chain, 2].tolist())}
    assert {5, 7, 11} <= set(listed)
    nparts = 1
    key12 = nparts | (1 << 24)
    sdesc, srec = plan.streams(nparts, walkers=12)
    steps_all = ei.shape[1] + N
    limit = 2.1 * steps_all / (nparts * 48)
    assert listed[5] < limit and listed[11] < limit < listed[7]                      # (what the graph was built for)
    chunks = _schedule_array(plan, key12, 3).view(-1, 4)
    n_chunk = int(_schedule_array(plan, key12, 0)[-1])
    chain12 = set(chunks[:n_chunk][(chunks[:n_chunk, 3] & 1) != 0, 0].tolist())        # rows with a CHUNK_FIRST chunk
    assert chain12 == {r for r, l in listed.items() if l > limit} and 7 in chain12
    markers = srec[srec[:, 1] == R, 0].tolist()
    assert len(markers) == len(set(markers)) and set(markers) == set(range(N)) - chain12
    assert {5, 11} <= set(markers)
    # inside every stream: row lengths ascending
    deg = torch.bincount(ei[0], minlength=N)
    for g_ in range(nparts * 64):
        begin, steps = sdesc[g_].tolist()
        rows = srec[begin:begin + steps][srec[begin:begin + steps, 1] == R, 0]
        lens = deg[rows.long()]
        assert bool((lens[1:] >= lens[:-1]).all())
    # the sixteen-walker schedule: every listed chain row is a chain row
    chunks16 = _schedule_array(plan, nparts, 3).view(-1, 4)
    n16 = int(_schedule_array(plan, nparts, 0)[-1])
    assert set(chunks16[:n16][(chunks16[:n16, 3] & 1) != 0, 0].tolist()) == set(listed)


def test_schedules_serve_every_row_once_on_random_graphs():
    """Schedule builder under random shapes (one node .. 2,000, no edges .. 30,000, hub rows of 257 .. 9,000 edges, 1 .. 32
    partitions, twelve and sixteen walkers): every row is either one chain row -- its chunks cover its edges -- or the row of
    exactly one stream marker; the streams' steps are their rows' edges + markers; the update waves' streams stay empty."""
    import random
    from ultra_amd.rspmm import Plan
    rng = random.Random(0)
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
        deg = torch.bincount(ei[0], minlength=N)
        for nparts in (1, 24):
            for walkers in (12, 16):
                key = nparts | ((1 << 24) if walkers == 12 else 0)
                sdesc, srec = plan.streams(nparts, walkers=walkers)
                cp = _schedule_array(plan, key, 0)
                ch = _schedule_array(plan, key, 3).view(-1, 4)[:int(cp[-1])]
                chain_rows = ch[(ch[:, 3] & 1) != 0, 0].tolist()
                covered = {}
                for row, _, cnt, _ in ch.tolist():
                    covered[row] = covered.get(row, 0) + cnt
                markers = srec[srec[:, 1] == R, 0].tolist()
                where = "graph %d, %d partitions, %d walkers" % (it, nparts, walkers)
                assert len(markers) == len(set(markers)) and len(chain_rows) == len(set(chain_rows)), where
                assert not (set(markers) & set(chain_rows)) and set(markers) | set(chain_rows) == set(range(N)), where
                assert all(covered[r] == int(deg[r]) for r in chain_rows), where
                assert int(sdesc[:, 1].sum()) == sum(int(deg[r]) + 1 for r in markers), where
                if walkers == 12:
                    assert int(sdesc[:, 1].view(nparts, 64)[:, 48:].sum()) == 0, where


def test_stream_work_follows_the_wave_age_shares():
    """plan.cpp WAVE_SHARE: a CU issues oldest wave first, so the schedule gives the four wave quartets of a workgroup
    1.7 / 1.3 / 0.7 / 0.3 of an even share of its stream steps (they then finish their walks together)."""
    from ultra_amd.rspmm import Plan
    ei, et = helpers.random_graph(num_node=6000, num_edge=90000, num_relation=12, seed=3)
    plan = Plan(ei, et, 6000, 12, exact_order=True)
    nparts = 2
    sdesc, _ = plan.streams(nparts)
    quartet = sdesc[:, 1].view(nparts, 4, 16).sum(dim=2).double()          # steps per (workgroup, wave quartet)
    share = quartet / quartet.mean(dim=1, keepdim=True)
    want = torch.tensor([1.7, 1.3, 0.7, 0.3], dtype=torch.double)
    assert torch.allclose(share, want.expand_as(share), atol=0.05), share
    per_stream = sdesc[:, 1].view(nparts, 4, 16).double()
    assert float((per_stream.max(dim=2)[0] - per_stream.min(dim=2)[0]).max()) <= 40      # streams of one quartet: balanced among themselves


def test_workgroup_row_lists_partition_the_rows_along_the_streams():
    """Work list of the update tail (plan.cpp build_schedule): workgroup q's list = its chain rows + the rows of its 64
    streams, ascending, padded with -1 to whole 32-row tiles; the lists partition the rows."""
    from ultra_amd.rspmm import Plan
    from ultra_amd import _lib
    ei, et = helpers.random_graph(num_node=300, num_edge=2000, num_relation=9, seed=8, hub=(11, 700))
    N, R = 300, 9
    plan = Plan(ei, et, N, R, exact_order=True)
    for nparts in (1, 4):
        sdesc, srec = plan.streams(nparts)
        prow, prow_ptr = plan.part_rows(nparts)
        assert prow_ptr.shape == (nparts + 1,) and int(prow_ptr[0]) == 0 and int(prow_ptr[-1]) == prow.numel()
        n_chain = plan.info()["n_chain_row"]
        chain_rows = set(plan.export(_lib.ARR_ITEM).view(-1, 4)[:n_chain, 0].tolist())
        assert len(chain_rows) >= 1
        seen = []
        for q in range(nparts):
            b, e = int(prow_ptr[q]), int(prow_ptr[q + 1])
            assert (e - b) % 32 == 0
            mine = prow[b:e]
            real = mine[mine >= 0]
            assert torch.equal(mine[:real.numel()], real) and real.numel() > e - b - 32     # padding at the end, under one tile
            assert torch.equal(real, real.sort()[0])
            stream_rows = set()
            for g in range(q * 64, (q + 1) * 64):
                first, steps = sdesc[g].tolist()
                seg = srec[first:first + steps]
                stream_rows |= set(seg[seg[:, 1] == R, 0].tolist())
            assert stream_rows <= set(real.tolist())
            assert set(real.tolist()) - stream_rows <= chain_rows
            seen += real.tolist()
        assert sorted(seen) == list(range(N))


def test_generated_assembly_header_is_in_sync_with_its_generator(tmp_path, monkeypatch):
    """csrc/rspmm_order_asm.hpp is generated (tools/gen_order_asm.py) and committed: the committed text must be what the
    generator writes with its default switches, so that the two can not drift apart."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in ("ULTRA_GEN_REC_POLICY", "ULTRA_GEN_OUT_POLICY", "ULTRA_GEN_PROD_NOWAIT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("ULTRA_GEN_OUT", str(tmp_path / "asm.hpp"))
    spec = importlib.util.spec_from_file_location("gen_order_asm", os.path.join(root, "tools", "gen_order_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.main()
    fresh = (tmp_path / "asm.hpp").read_text()
    committed = open(os.path.join(root, "ultra_amd", "csrc", "rspmm_order_asm.hpp")).read()
    assert fresh == committed
    # every path out of a statement drains the vector-memory queue, and the statements declare what they clobber
    # 2 x 3 sums x 2 messages (stream walk: plain, and with the finished rows parked in LDS for the update waves) + 2 messages
    # (chain producers)
    assert fresh.count("asm volatile(") == 14 and fresh.count('"memory"') == 14
    for block in fresh.split("asm volatile(")[1:]:
        assert "s_waitcnt vmcnt(0)" in block.split(");")[0]
    # ... and the generator of the relation-graph layer's measurement build still runs (its header is not part of the tree)
    for k in ("ULTRA_GEN_DOL_NS", "ULTRA_GEN_DOL_TOUCH"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("ULTRA_GEN_DENSE_OUT", str(tmp_path / "dense_asm.hpp"))
    spec = importlib.util.spec_from_file_location("gen_dense_order_asm", os.path.join(root, "tools", "gen_dense_order_asm.py"))
    gen2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen2)
    gen2.main()
    assert "v_mfma_f32_16x16x4" in (tmp_path / "dense_asm.hpp").read_text()
