"""Evaluation / sampling glue around the hot path (the reference's ultra/tasks.py), pure torch on
whatever device the graph lives on.  Same names, arguments and results as the reference:

  edge_match             tasks.py:7-39     all graph edges matching each query key, via mixed-radix keys + searchsorted
  negative_sampling      tasks.py:42-76
  prefetch_negatives     (the same, one batch ahead of the training step on a side stream; script/run.py:53-55)
  all_negative           tasks.py:79-91    (bs, 3) positives -> (bs, N, 3) tail- and head-candidate batches
  strict_negative_mask   tasks.py:94-130   filtered-ranking masks
  compute_ranking        tasks.py:133-141  rank = #(masked candidates scoring >= positive) + 1
  build_relation_graph   tasks.py:144-199  relation-relation graph with the 4 fundamental interactions
"""
import torch

from .data import Data


def _mixed_radix_scale(edge_index):
    # one radix per row: (max + 1); key = sum_i index[i] * prod_{j>i} radix[j]
    radix = edge_index.max(dim=1)[0] + 1
    total = 1
    for r in radix.tolist():
        total *= int(r)
    assert total < torch.iinfo(torch.long).max, "edge key overflows int64"
    scale = torch.ones_like(radix)
    for i in range(len(radix) - 2, -1, -1):
        scale[i] = scale[i + 1] * radix[i + 1]
    return scale


class EdgeKeyIndex(object):
    """The sorted mixed-radix keys of a static edge list: edge_match sorts the whole graph on every call
    (tasks.py:25-26); for the evaluation loop that sort is hoisted and reused across batches."""

    def __init__(self, edge_index):
        self.scale = _mixed_radix_scale(edge_index).unsqueeze(-1)
        self.edge_key, self.order = (edge_index * self.scale).sum(dim=0).sort()


_KEY_INDEX_CACHE = {}


def _key_index(tag, tensors, build):
    key = (tag,) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)
    hit = _KEY_INDEX_CACHE.get(key)
    if hit is None:
        if len(_KEY_INDEX_CACHE) > 16:
            _KEY_INDEX_CACHE.clear()
        hit = (build(), tensors)          # keep the tensors alive so a recycled data_ptr cannot alias
        _KEY_INDEX_CACHE[key] = hit
    return hit[0]


def edge_match(edge_index, query_index, index=None):
    """For every query column, the ids of all graph edges with the same key.
    Returns (edge ids concatenated query by query, matches per query)."""
    if index is None:
        index = EdgeKeyIndex(edge_index)
    scale, edge_key, order = index.scale, index.edge_key, index.order
    query_key = (query_index * scale).sum(dim=0)
    lo = torch.searchsorted(edge_key, query_key, right=False)
    hi = torch.searchsorted(edge_key, query_key, right=True)
    num_match = hi - lo
    # ranges [lo, hi) flattened: position p of query q is lo[q] + (p - first position of q)
    first = num_match.cumsum(0) - num_match
    total = int(num_match.sum())
    pos = torch.arange(total, device=query_index.device)
    pos = pos + (lo - first).repeat_interleave(num_match)
    return order[pos], num_match


def strict_negative_mask(data, batch):
    """(t_mask, h_mask): True where a candidate entity is a valid negative (not a known true answer)."""
    pos_h, pos_t, pos_r = batch.t()
    masks = []
    # tails of every (h, r) in the graph, then heads of every (t, r)
    for known, anchor, answer_row, positive in ((0, pos_h, 1, pos_t), (1, pos_t, 0, pos_h)):
        query = torch.stack([anchor, pos_r])
        # the sorted key index of the static graph is built once and reused by every batch (the reference re-sorts the
        # whole edge list inside each edge_match call, tasks.py:25-26)
        index = _key_index("strict%d" % known, (data.edge_index, data.edge_type),
                           lambda known=known: EdgeKeyIndex(torch.stack([data.edge_index[known], data.edge_type])))
        edge_id, count = edge_match(None, query, index=index)
        truth = data.edge_index[answer_row, edge_id]
        sample = torch.arange(len(count), device=batch.device).repeat_interleave(count)
        mask = torch.ones(len(count), data.num_nodes, dtype=torch.bool, device=batch.device)
        mask[sample, truth] = False
        mask.scatter_(1, positive.unsqueeze(-1), False)
        masks.append(mask)
    return masks[0], masks[1]


STRICT_SAMPLER_KERNEL = True      # (A/B switch for tests: the masks + nonzero() formulation below is the other side)


def _answer_keys(data, known):
    """The graph's distinct (anchor, relation, answer) triples as ascending int64 keys (anchor * R + relation) * N + answer --
    known = 0: tails of every (head, relation); 1: heads of every (tail, relation).  Built once per graph."""
    def build():
        n, r = int(data.num_nodes), int(data.num_relations)
        assert n * n * max(r, 1) < 2 ** 63, "edge key overflows int64"          # (tasks.py:19, same bound)
        anchor, answer = data.edge_index[known], data.edge_index[1 - known]
        return torch.unique((anchor * r + data.edge_type) * n + answer).contiguous()
    return _key_index("answers%d" % known, (data.edge_index, data.edge_type), build)


def _strict_negatives_gpu(data, anchor, relation, positive, num_negative, known, rand=None):
    """num_negative strict negatives per positive through csrc/sampling.hip: the reference's picks
    candidate[floor(rand * count)] (tasks.py:57-61) for the same torch.rand draws, no mask, no host synchronisation.
    rand: the (rows, num_negative) uniform draws to use instead of torch.rand on the device (tests feed the CPU generator's, to
    replay batches recorded from the reference)."""
    import ctypes
    from ._lib import check, lib
    keys = _answer_keys(data, known)
    rows = len(anchor)
    if rand is None:
        rand = torch.rand(rows, num_negative, device=anchor.device)
    rand = rand.to(device=anchor.device, dtype=torch.float32).contiguous()
    out = torch.empty(rows, num_negative, dtype=torch.long, device=anchor.device)
    anchor, relation, positive = anchor.contiguous(), relation.contiguous(), positive.contiguous()
    check(lib.ultra_strict_negatives(keys.data_ptr(), keys.numel(), anchor.data_ptr(), relation.data_ptr(), positive.data_ptr(),
                                     rand.data_ptr(), rows, num_negative, int(data.num_nodes), int(data.num_relations),
                                     out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(anchor.device).cuda_stream)))
    return out


def negative_sampling(data, batch, num_negative, strict=True):
    batch_size = len(batch)
    pos_h, pos_t, pos_r = batch.t()
    half = batch_size // 2
    if strict and STRICT_SAMPLER_KERNEL and batch.is_cuda and batch.dtype == torch.long and data.edge_index.is_cuda \
            and data.edge_index.dtype == torch.long and torch.get_default_dtype() == torch.float32:
        # the same draws, in the same order, as the masks + nonzero() formulation below (tails of the first half, then heads of
        # the second) -- two small kernels and no host synchronisation instead of ~ 60 launches and three
        neg_t = _strict_negatives_gpu(data, pos_h[:half], pos_r[:half], pos_t[:half], num_negative, known=0)
        neg_h = _strict_negatives_gpu(data, pos_t[half:], pos_r[half:], pos_h[half:], num_negative, known=1)
    elif strict:
        t_mask, h_mask = strict_negative_mask(data, batch)

        def draw(mask):
            candidate = mask.nonzero()[:, 1]
            count = mask.sum(dim=-1)
            rand = torch.rand(len(mask), num_negative, device=batch.device)
            index = (rand * count.unsqueeze(-1)).long() + (count.cumsum(0) - count).unsqueeze(-1)
            return candidate[index]

        neg_t = draw(t_mask[:half])
        neg_h = draw(h_mask[half:])
    else:
        neg = torch.randint(data.num_nodes, (batch_size, num_negative), device=batch.device)
        neg_t, neg_h = neg[:half], neg[half:]
    h_index = pos_h.unsqueeze(-1).repeat(1, num_negative + 1)
    t_index = pos_t.unsqueeze(-1).repeat(1, num_negative + 1)
    r_index = pos_r.unsqueeze(-1).repeat(1, num_negative + 1)
    t_index[:half, 1:] = neg_t
    h_index[half:, 1:] = neg_h
    return torch.stack([h_index, t_index, r_index], dim=-1)


# measurements: append {"open": True} before creating a prefetch_negatives generator and it fills in what its stream check saw
PREFETCH_REPORT = []


def overlapping_stream(dev, candidates=8, busy_cycles=2_000_000):
    """A stream whose kernels run BESIDE those of the current stream of `dev`, found by trying: the runtime maps every stream onto
    one of a handful of hardware queues when it is created, and a stream that shares the current stream's queue runs behind its
    kernels, not beside them -- high priority does not prevent that.  (Measured: the captured fine-tuning step takes 2.9 ms with the
    sampler on a high-priority stream made first in the process and 3.25 -- the step plus the sampler, end to end -- with the same
    stream made after bench.py's pipelined forward had made its own; profiles/r6_experiments.txt.)  Each candidate (high and normal
    priority in turn) gets one tiny kernel while the current stream spins for ~ 1 ms; the first whose kernel finishes before the
    spin does is taken."""
    main = torch.cuda.current_stream(dev)
    probe = torch.zeros(64, device=dev)
    best = None
    with torch.cuda.device(dev):
        for i in range(int(candidates)):
            cand = torch.cuda.Stream(priority=-1 if i % 2 == 0 else 0)
            best = best or cand
            if not hasattr(torch.cuda, "_sleep"):
                break
            cand.wait_stream(main)
            torch.cuda.synchronize(dev)
            done_main, done_cand = torch.cuda.Event(), torch.cuda.Event()
            torch.cuda._sleep(int(busy_cycles))
            done_main.record(main)
            with torch.cuda.stream(cand):
                probe.add_(1.0)
                done_cand.record(cand)
            done_cand.synchronize()
            beside = not done_main.query()
            torch.cuda.synchronize(dev)
            if beside:
                return cand
    return best


def prefetch_negatives(batches, data, num_negative, strict=True):
    """negative_sampling() over an iterable of positive batches, one batch AHEAD of the training step and on a side stream:

        for batch in prefetch_negatives(loader, train_data, num_negative, strict):      # script/run.py:53-55
            pred = model(train_data, batch) ...

    The strict sampler is ~ 60 small launches and three host synchronisations (nonzero, the match counts).  Issued on
    the training stream it waits for the previous step's backward and optimiser to drain, then the GPU idles while the
    host reads the counts back; issued here -- after the previous step's launches are enqueued, on its own stream --
    its kernels run beside that step's backward and the host synchronises with the side stream alone.  The draws come
    from the default generator in the same order as the plain loop makes them.

    The positives are PRODUCED on the side stream too: `next(batches)` runs under it, so whatever kernels the iterable
    launches to build a batch -- a DataLoader over a GPU tensor stacks its rows with a kernel on the current stream
    (script/run.py:32-34 keeps train_triplets on the device) -- are ordered in front of the sampler that reads the batch,
    instead of behind the previous step's backward on the training stream (where the sampler would read the batch before it
    is written: ADVICE r5).  What the iterable READS must be complete when the loop starts (the triple list is static).
    A graph on the CPU: the plain loop, unchanged."""
    it = iter(batches)
    dev = data.edge_index.device
    if dev.type != "cuda":
        for batch in it:
            yield negative_sampling(data, batch, num_negative, strict=strict)
        return
    with torch.cuda.device(dev):
        side = overlapping_stream(dev)
        # the triple list and the graph were written on the caller's stream before this loop began
        side.wait_stream(torch.cuda.current_stream(dev))

    def sample():
        with torch.cuda.stream(side):
            batch = next(it)            # (StopIteration passes through)
            return negative_sampling(data, batch.to(dev, non_blocking=True), num_negative, strict=strict)

    try:
        ahead = sample()
    except StopIteration:
        return
    # The stream is re-checked while the loop runs: what the caller launches between two draws may use hardware queues of its own
    # (a hipGraph replay does), so the one-off trial above cannot see every collision.  An event behind the caller's step and one
    # behind the draw issued right after it are compared: a draw that ends AFTER the step it was issued beside ran behind it.  Twice
    # in a row -> the next candidate stream (at most six times; a loop whose steps are shorter than a draw ends up on the last one,
    # which costs nothing).  The first observations are WAITED for (the host would otherwise be dozens of replays ahead before an
    # event completes: a few steps of a warm-up run in lock-step with the GPU, until two draws in a row end in time); later ones are
    # looked at when they happen to be complete.
    waited_left, checks_left, late_in_a_row, fine_in_a_row, switches_left, pending = 12, 24, 0, 0, 6, []
    report = PREFETCH_REPORT[-1] if PREFETCH_REPORT and PREFETCH_REPORT[-1].get("open") else {}
    report.update({"observed": 0, "late": 0, "switches": 0, "waited": 0})
    while ahead is not None:
        main = torch.cuda.current_stream(dev)
        main.wait_stream(side)
        ahead.record_stream(main)        # (allocated on the side stream, read by the step on the training stream)
        current, ahead = ahead, None
        yield current                    # the caller enqueues its step ...
        behind_step = None
        if checks_left > 0 and switches_left > 0 and not torch.cuda.is_current_stream_capturing():
            behind_step = torch.cuda.Event(enable_timing=True)
            behind_step.record(main)
        try:
            ahead = sample()             # ... and the next batch's negatives are drawn while the GPU works through it
        except StopIteration:
            ahead = None
        if behind_step is not None:
            behind_draw = torch.cuda.Event(enable_timing=True)
            behind_draw.record(side)
            pending.append((behind_step, behind_draw))
            checks_left -= 1
            if waited_left > 0 and fine_in_a_row < 2:
                waited_left -= 1
                report["waited"] += 1
                behind_step.synchronize()
                behind_draw.synchronize()
        while pending and pending[0][0].query() and pending[0][1].query():
            step_end, draw_end = pending.pop(0)
            late = step_end.elapsed_time(draw_end) > 0.0
            report["observed"] += 1
            report["late"] += int(late)
            late_in_a_row, fine_in_a_row = (late_in_a_row + 1, 0) if late else (0, fine_in_a_row + 1)
            if late_in_a_row >= 2 and switches_left > 0:
                with torch.cuda.device(dev):
                    fresh = torch.cuda.Stream(priority=0 if switches_left % 2 else -1)
                fresh.wait_stream(side)
                side, late_in_a_row, fine_in_a_row, switches_left, pending = fresh, 0, 0, switches_left - 1, []
                report["switches"] += 1
                waited_left, checks_left = max(waited_left, 4), 24


def all_negative(data, batch):
    pos_h, pos_t, pos_r = batch.t()
    n = data.num_nodes
    every = torch.arange(n, device=batch.device).unsqueeze(0).expand(len(batch), -1)
    r = pos_r.unsqueeze(-1).expand(-1, n)
    t_batch = torch.stack([pos_h.unsqueeze(-1).expand(-1, n), every, r], dim=-1)
    h_batch = torch.stack([every, pos_t.unsqueeze(-1).expand(-1, n), r], dim=-1)
    return t_batch, h_batch


def compute_ranking(pred, target, mask=None):
    pos_pred = pred.gather(-1, target.unsqueeze(-1))
    worse_or_equal = pos_pred <= pred      # ties count against the positive (tasks.py:137)
    if mask is not None:
        worse_or_equal = worse_or_equal & mask
    return worse_or_equal.sum(dim=-1) + 1


def known_answers(data, batch, mode="tail"):
    """Ragged list of the known true tails (mode="tail": of every (h, r)) or heads (mode="head": of every
    (t, r)) per query, de-duplicated and including the positive itself -- what strict_negative_mask
    (tasks.py:94-130) zeroes, without building the (batch, N) mask.  Returns (ptr (bs + 1), index)."""
    pos_h, pos_t, pos_r = batch.t()
    if mode == "tail":
        keyed, anchor, answer_row, positive = 0, pos_h, 1, pos_t
    else:
        keyed, anchor, answer_row, positive = 1, pos_t, 0, pos_h
    idx = _key_index(mode, (data.edge_index, data.edge_type),
                     lambda: EdgeKeyIndex(torch.stack([data.edge_index[keyed], data.edge_type])))
    edge_id, count = edge_match(None, torch.stack([anchor, pos_r]), index=idx)
    truth = data.edge_index[answer_row, edge_id]
    sample = torch.arange(len(count), device=batch.device).repeat_interleave(count)
    n = data.num_nodes
    key = torch.cat([sample * n + truth, torch.arange(len(batch), device=batch.device) * n + positive])
    key = torch.unique(key)                                    # sorted: grouped by query, ids ascending
    ptr = torch.searchsorted(key, torch.arange(len(batch) + 1, device=batch.device) * n)
    return ptr, key % n


def filtered_ranking(data, batch, pred, mode="tail"):
    """(ranking, num_negative) of the positives under the filtered protocol == compute_ranking(pred, pos,
    strict_negative_mask(...)) and mask.sum(-1), through the fused HIP kernel (no (batch, N) mask)."""
    import ctypes
    from ._lib import check, lib
    if not pred.is_cuda:
        raise RuntimeError("ultra_amd.tasks.filtered_ranking: expected a GPU tensor; the MI355X engine has no CPU path")
    pos = (batch[:, 1] if mode == "tail" else batch[:, 0]).contiguous()
    ptr, index = known_answers(data, batch, mode)
    ptr, index = ptr.contiguous(), index.contiguous()      # (referenced until the launch is enqueued)
    pred = pred.float().contiguous()
    rank = torch.empty(len(batch), dtype=torch.long, device=pred.device)
    num_neg = torch.empty_like(rank)
    check(lib.ultra_filtered_rank(pred.data_ptr(), pos.data_ptr(), ptr.data_ptr(), index.data_ptr(),
                                  pred.shape[0], pred.shape[1], rank.data_ptr(), num_neg.data_ptr(),
                                  ctypes.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)))
    return rank, num_neg


def filtered_ranking_masks(data, batch, pred, mode="tail"):
    """filtered_ranking() in the reference's own formulation -- strict_negative_mask + compute_ranking
    (tasks.py:94-141) -- for tensors on any device; the fused kernel is tested against it."""
    t_mask, h_mask = strict_negative_mask(data, batch)
    mask = t_mask if mode == "tail" else h_mask
    pos = batch[:, 1] if mode == "tail" else batch[:, 0]
    return compute_ranking(pred, pos, mask), mask.sum(dim=-1)


def relation_graph_bits(graph):
    """(adj, row_counts): the four adjacency bit matrices [hh | tt | ht | th] of graph's relation graph, shape
    (4, num_relations, W) int32 words, and the edges per (type, row) -- built by the HIP kernels of csrc/relgraph.hip."""
    import ctypes
    from ._lib import check, lib
    ei, et = graph.edge_index.to(torch.int64).contiguous(), graph.edge_type.to(torch.int64).contiguous()
    n, r = int(graph.num_nodes), int(graph.num_relations)
    w = (r + 31) // 32
    dev = ei.device
    hbits = torch.zeros(n * w, dtype=torch.int32, device=dev)
    tbits = torch.zeros(n * w, dtype=torch.int32, device=dev)
    adj = torch.zeros(4, r, w, dtype=torch.int32, device=dev)
    counts = torch.empty(4 * r, dtype=torch.int64, device=dev)
    check(lib.ultra_relation_graph_bits(ei.data_ptr(), et.data_ptr(), ei.shape[1], n, r, hbits.data_ptr(), tbits.data_ptr(),
                                        adj.data_ptr(), counts.data_ptr(),
                                        ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return adj, counts


def build_relation_graph_gpu(graph):
    """build_relation_graph for a graph resident on the GPU: bit-matrix kernels instead of the reference's four sparse
    products (tasks.py:186-189); relation_graph.edge_index / edge_type equal the reference's element for element.  The
    bit matrices stay attached (relation_graph.adjacency_bits) for consumers that want plan format without the edge list."""
    import ctypes
    from ._lib import check, lib
    adj, counts = relation_graph_bits(graph)
    r = int(graph.num_relations)
    offsets = torch.cumsum(counts, 0) - counts
    total = int(counts.sum())
    dev = adj.device
    edge_index = torch.empty(2, total, dtype=torch.int64, device=dev)
    edge_type = torch.empty(total, dtype=torch.int64, device=dev)
    check(lib.ultra_relation_graph_emit(adj.data_ptr(), offsets.data_ptr(), r, total, edge_index.data_ptr(), edge_type.data_ptr(),
                                        ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    graph.relation_graph = Data(edge_index=edge_index, edge_type=edge_type, num_nodes=r, num_relations=4, adjacency_bits=adj)
    return graph


def relation_graph_dense_adjacency(adj):
    """The byte adjacency of the reference-order layer kernel (plan.hpp `a_ex`) from the bit matrices, on the device."""
    import ctypes
    from ._lib import check, lib
    r = adj.shape[1]
    nt = (r + 15) // 16
    out = torch.empty(nt * nt * 1024, dtype=torch.uint8, device=adj.device)
    check(lib.ultra_relation_graph_dense_adjacency(adj.contiguous().data_ptr(), r, out.data_ptr(),
                                                   ctypes.c_void_p(torch.cuda.current_stream(adj.device).cuda_stream)))
    return out


def build_relation_graph(graph, node_chunk=1 << 16):
    """Relation graph of a KG that already contains inverse edges: nodes are relation ids, an edge
    (r1, r2) of type hh / tt / ht / th exists iff some entity is a head (h) or tail (t) of r1 and of
    r2 respectively.  The reference forms four sparse products E^T E (tasks.py:152-189) and keeps
    only their sparsity pattern; here the pattern is accumulated from dense per-chunk incidence
    GEMMs.  Edge order matches the reference: hh, tt, ht, th blocks, each row-major sorted."""
    edge_index, edge_type = graph.edge_index, graph.edge_type
    num_nodes, num_rels = graph.num_nodes, graph.num_relations
    device = edge_index.device
    if edge_index.is_cuda:
        return build_relation_graph_gpu(graph)
    counts = [torch.zeros(num_rels, num_rels, device=device) for _ in range(4)]
    key_h = torch.unique(edge_index[0] * num_rels + edge_type)
    key_t = torch.unique(edge_index[1] * num_rels + edge_type)
    for lo in range(0, num_nodes, node_chunk):
        hi = min(num_nodes, lo + node_chunk)
        inc = []
        for key in (key_h, key_t):
            sel = key[(key >= lo * num_rels) & (key < hi * num_rels)] - lo * num_rels
            m = torch.zeros((hi - lo) * num_rels, device=device)
            m[sel] = 1.0
            inc.append(m.view(hi - lo, num_rels))
        bh, bt = inc
        counts[0] += bh.t() @ bh
        counts[1] += bt.t() @ bt
        counts[2] += bh.t() @ bt
        counts[3] += bt.t() @ bh
    blocks, types = [], []
    for k, c in enumerate(counts):
        idx = (c > 0).nonzero().t()
        blocks.append(idx)
        types.append(torch.full((idx.shape[1],), k, dtype=torch.long, device=device))
    rel_graph = Data(edge_index=torch.cat(blocks, dim=1), edge_type=torch.cat(types), num_nodes=num_rels,
                     num_relations=4)
    graph.relation_graph = rel_graph
    return graph
