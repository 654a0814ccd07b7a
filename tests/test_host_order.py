"""ultra_amd/host_order.py on the CPU: the summation tree of the host BLAS for nn.Linear(128, 1) is recovered by probing,
parsed into the stage program the readout kernel executes, and the program's numpy restatement reproduces torch."""
import numpy as np
import pytest
import torch

from ultra_amd import host_order as ho


def chain(elems, base=None, fused=True):
    node = base
    for k in elems:
        node = k if node is None else (("f" if fused else "a"), node, k)
    return node


def fold(vals):
    vals = list(vals)
    half = len(vals) // 2
    while half >= 1:
        for p in range(half):
            a, b = vals[p], vals[p + half]
            vals[p] = a if b is None else ("a", a, b)
        half //= 2
    return vals[0]


def eval_tree(t, x, w):
    """fp32 evaluation of an annotated tree: leaf = rounded product, 'f' = fma onto the left value, 'a' = add."""
    def r32(a):
        return a.astype(np.float32).astype(np.float64)
    if isinstance(t, int):
        return r32(x[:, t] * w[t])
    if t[0] == "f":
        return r32(x[:, t[2]] * w[t[2]] + eval_tree(t[1], x, w))
    return r32(eval_tree(t[1], x, w) + eval_tree(t[2], x, w))


def random_operands(rows=300, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(rows, 128, generator=g).numpy().astype(np.float64), torch.randn(128, generator=g).numpy().astype(np.float64)


def test_parse_four_unfused_strided_chains_with_a_scalar_prologue():
    """The shape seen on an AMD EPYC: elements 0-3 run into lane 0, then four chains with stride 4 whose products are
    rounded before they are added (no fma), folded (0 + 2) + (1 + 3)."""
    lanes = [[0, 1, 2, 3] + list(range(4, 128, 4)), list(range(5, 128, 4)), list(range(6, 128, 4)), list(range(7, 128, 4))]
    tree = ("a", ("a", chain(lanes[0], fused=False), chain(lanes[2], fused=False)),
            ("a", chain(lanes[1], fused=False), chain(lanes[3], fused=False)))
    stages = ho.tree_to_stages(tree)
    assert len(stages) == 1 and stages[0][0] == 4 and not stages[0][1]
    assert [[k & 255 for k in lane] for lane in stages[0][2]] == lanes
    assert all(k & ho.UNFUSED for lane in stages[0][2] for k in lane[1:])
    x, w = random_operands()
    assert np.array_equal(ho.emulate(stages, x, w), eval_tree(tree, x, w).astype(np.float32))
    prog = ho.stages_to_program(stages)
    assert prog[:5] == [1, 4, 0, 12, 5] and prog[5:7] == [52, 4] and len(prog) == 12 + 40 + 3 * 32
    assert prog[12:16] == [0, 1 + ho.UNFUSED, 2 + ho.UNFUSED, 3 + ho.UNFUSED] and prog[12 + 35] == ho.PAD


def test_parse_sixteen_fused_lanes_plus_a_masked_tail():
    """The shape seen on an Intel Xeon (AVX-512): 16 fma lanes over k = 1..112 (element 0 first in lane 0), reduced to a
    scalar onto which product 113 is fused; the other 14 tail products are rounded and folded in."""
    lanes = [([0] if p == 0 else []) + list(range(p + 1, 113, 16)) for p in range(16)]
    main = fold([chain(l) for l in lanes])
    tree = fold([chain([113], main)] + [113 + i for i in range(1, 15)] + [None])
    stages = ho.tree_to_stages(tree)
    assert len(stages) == 2 and stages[0] == (16, False, lanes) and stages[1][1] is True
    assert sorted(k & 255 for _, _, ls in stages for l in ls for k in l) == list(range(128))
    x, w = random_operands(seed=1)
    assert np.array_equal(ho.emulate(stages, x, w), eval_tree(tree, x, w).astype(np.float32))


def test_rejects_trees_outside_the_family():
    with pytest.raises(ValueError):       # a carried value in a lane other than 0
        ho.tree_to_stages(("a", ("f", ("a", ("f", 0, 1), ("f", 2, 3)), 4), ("f", ("a", ("f", 5, 6), ("f", 7, 8)), 9)))


def test_emulation_of_the_sequential_chain():
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(50, 128, generator=g), torch.randn(128, generator=g)
    acc = torch.zeros(50, dtype=torch.float64)
    for k in range(128):
        acc = (x[:, k].double() * w[k].double() + acc).float().double()
    assert np.array_equal(ho.emulate(ho.sequential_stages(128), x.numpy(), w.numpy()), acc.float().numpy())


def test_host_blas_order_is_recovered_and_reproduces_torch(monkeypatch):
    monkeypatch.delenv("ULTRA_READOUT_ORDER", raising=False)
    ho._CACHE.clear()
    stages, source = ho.readout_stages(128)
    if not source.startswith("host BLAS"):
        pytest.skip("this host's BLAS sums nn.Linear(128, 1) outside the lanes-and-fold family: the ascending chain is used")
    g = torch.Generator().manual_seed(1)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 4))      # few shares: few remainder rows (see below)
    for rows in (4096, 116328):
        x, w, b = torch.randn(rows, 128, generator=g), torch.randn(1, 128, generator=g), torch.randn(1, generator=g)
        want = torch.nn.functional.linear(x, w, b)[:, 0].numpy()
        got = (ho.emulate(stages, x.numpy(), w[0].numpy()).astype(np.float64) + float(b)).astype(np.float32)
        # the BLAS sums the last few rows of every thread's share with a remainder kernel: everything else is bit-equal
        assert (got == want).mean() >= 0.995 and np.abs(got - want).max() <= 4e-6 * np.abs(want).max()
    torch.set_num_threads(threads)
    prog, _ = ho.readout_program(128)
    assert prog[0] == len(stages) and len(prog) <= 640


def test_sequential_order_on_request(monkeypatch):
    monkeypatch.setenv("ULTRA_READOUT_ORDER", "sequential")
    ho._CACHE.clear()
    stages, source = ho.readout_stages(128)
    assert stages == ho.sequential_stages(128) and source == "sequential"
    ho._CACHE.clear()


def test_order_file_round_trip_cache_and_id(monkeypatch, tmp_path):
    """The order in use is named (order_id), cached on disk per host, and can be carried to another host as a file."""
    monkeypatch.setenv("ULTRA_ORDER_CACHE_DIR", str(tmp_path / "cache"))
    monkeypatch.delenv("ULTRA_READOUT_ORDER", raising=False)
    ho._CACHE.clear()
    threads = torch.get_num_threads()
    stages, source = ho.readout_stages(128)          # probed in a helper process: this process's thread count is untouched
    assert torch.get_num_threads() == threads
    files = list((tmp_path / "cache").glob("readout_order_*.json")) if source.startswith("host BLAS") else []
    if source.startswith("host BLAS"):
        assert len(files) == 1
        ho._CACHE.clear()
        again, source2 = ho.readout_stages(128)       # second process start: read from the cache, no probe
        assert again == stages and source2.endswith("[cached]")
    oid = ho.order_id(stages)
    assert oid.startswith("order-") and len(oid) == 14 and ho.describe(128).startswith(oid)
    path = tmp_path / "order.json"
    ho.save_stages(str(path), stages, source)
    monkeypatch.setenv("ULTRA_READOUT_ORDER", str(path))
    ho._CACHE.clear()
    loaded, source3 = ho.readout_stages(128)
    assert loaded == stages and ho.order_id(loaded) == oid and source3.startswith("file order.json")
    assert ho.order_id(ho.sequential_stages(128)) != oid or stages == ho.sequential_stages(128)
    ho._CACHE.clear()


def test_unusable_orders_fall_back_to_the_sequential_chain(monkeypatch, tmp_path):
    """A missing / malformed order file, or a legal tree whose program exceeds the kernel's 640 words, must not break
    the forward: sequential chain + warning."""
    monkeypatch.setenv("ULTRA_READOUT_ORDER", str(tmp_path / "nope.json"))
    ho._CACHE.clear()
    with pytest.warns(UserWarning):
        stages, source = ho.readout_stages(128)
    assert stages == ho.sequential_stages(128) and source.startswith("sequential")
    # eight stages of sixteen lanes: legal, but 1 + 8 (2 + 32) + 128 lanes padded to groups of eight = far beyond 640 words
    big = [(16, s > 0, [[16 * s + p] for p in range(16)]) for s in range(8)]
    assert len(ho.stages_to_program(big)) > ho.PROGRAM_MAX_WORDS
    with pytest.raises(ValueError):
        ho._check_stages(big, 128)
    ho._CACHE.clear()
