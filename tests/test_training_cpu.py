"""Host-side restatements that the fine-tuning path's GPU kernels are tested against (tests/test_training_gpu.py), pinned
here -- on the CPU -- to torch autograd and to the reference's mask formulation."""
import torch

from ultra_amd import rspmm, synthetic, tasks


def test_first_layer_backward_restatement_matches_autograd():
    """rspmm._onehot_backward_torch (the padded-table formulation csrc/onehot_bwd.hip is compared with) against fp64 autograd
    of the dense rspmm on a one-hot input + boundary: relation gradient and values gradient, with a keep mask, a hub source
    and a source without out-edges; rspmm.out_edge_csr's layout on the way."""
    gen = torch.Generator().manual_seed(3)
    n, e, bs, num_rel = 60, 900, 4, 5
    ei = torch.randint(1, n, (2, e), generator=gen)
    ei[1, :300] = 7
    et = torch.randint(0, num_rel, (e,), generator=gen)
    rows = torch.tensor([7, 0, 33, 7])
    keep = (torch.rand(e, generator=gen) > 0.2).double()
    ptr, order, max_deg = rspmm.out_edge_csr(ei, et, n)
    assert ptr[-1].item() == e and max_deg == int(torch.bincount(ei[1], minlength=n).max())
    src_sorted, type_sorted = ei[1][order], et[order]
    assert (src_sorted.diff() >= 0).all()                                    # grouped by source ...
    same = src_sorted[1:] == src_sorted[:-1]
    assert (type_sorted[1:][same] >= type_sorted[:-1][same]).all()           # ... every source's edges by type
    values = torch.randn(bs, 16, generator=gen, dtype=torch.float64).requires_grad_()
    rel = torch.randn(bs, num_rel, 16, generator=gen, dtype=torch.float64).requires_grad_()
    og = torch.randn(bs, n, 16, generator=gen, dtype=torch.float64)
    x0 = torch.zeros(bs, n, 16, dtype=torch.float64).index_put((torch.arange(bs), rows), values)
    msg = rel[:, et] * x0[:, ei[1]] * keep.view(1, -1, 1)
    out = torch.zeros(bs, n, 16, dtype=torch.float64).index_add(1, ei[0], msg) + x0
    out.backward(og)
    got = rspmm._onehot_backward_torch(ptr, order, max_deg, ei, et, keep, rel.detach(), rows, values.detach(), og, True, True)
    torch.testing.assert_close(got[0], rel.grad, rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(got[1], values.grad, rtol=1e-10, atol=1e-10)


def test_strict_sampler_selection_rule_equals_the_mask_formulation():
    """csrc/sampling.hip's rule, restated in Python -- the idx-th entity id that is neither a known answer nor the positive,
    found by bisection against the query's slice of the sorted distinct answer keys (tasks._answer_keys) -- picks exactly
    candidate[idx] of the reference's masks + nonzero() formulation (tasks.py:57-61), for every idx of every query."""
    data = synthetic.make_kg(num_node=50, num_triple=600, num_relation_base=3, num_test=8, seed=2, relation_graph=False)
    n, r = int(data.num_nodes), int(data.num_relations)
    batch = torch.stack([data.edge_index[0, :6], data.edge_index[1, :6], data.edge_type[:6]], dim=-1).clone()
    batch[-1, 1] = (batch[-1, 1] + 11) % n
    t_mask, h_mask = tasks.strict_negative_mask(data, batch)
    for known, mask, anchor, positive in ((0, t_mask, batch[:, 0], batch[:, 1]), (1, h_mask, batch[:, 1], batch[:, 0])):
        keys = tasks._answer_keys(data, known).tolist()
        assert keys == sorted(set(keys))
        for q in range(len(batch)):
            base = (int(anchor[q]) * r + int(batch[q, 2])) * n
            known_ids = [k - base for k in keys if base <= k < base + n]
            pos = int(positive[q])
            excluded = sorted(set(known_ids) | {pos})
            candidate = mask[q].nonzero().flatten().tolist()
            assert len(candidate) == n - len(excluded)
            for idx in range(len(candidate)):
                a, b = idx, n - 1
                while a < b:
                    mid = (a + b) // 2
                    if mid + 1 - sum(1 for v in excluded if v <= mid) >= idx + 1:
                        b = mid
                    else:
                        a = mid + 1
                assert a == candidate[idx]
