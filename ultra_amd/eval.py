"""Filtered-ranking evaluation, query-sharded over the GPUs of a node (reference: script/run.py:121-226).

Same protocol as the reference's test(): for every test triple score all tails and all heads
(tasks.all_negative), mask known true answers (tasks.strict_negative_mask), rank the positive
(tasks.compute_ranking), then mr / mrr / hits@k (+ the unbiased hits@k_N estimator) over both
directions, plus the tail-only variants ("mrr-tail", ...).

What differs from the reference:
  * sharding: contiguous balanced shards of the test triples (distributed.shard_range) instead of
    DistributedSampler (which pads by repeating samples and therefore double-counts a few of them);
  * collectives: ONE all-gather of (ranking, num_negative, is_tail) rows per evaluation instead of six
    zero-padded all_reduce(SUM) calls (run.py:166-186) -- RCCL over xGMI on GPUs, gloo in the CPU tests.
"""
import math

import os

import torch

from . import distributed as udist
from . import models, tasks


def metrics_from_rankings(ranking, num_negative, metric_names):
    out = {}
    for metric in metric_names:
        name = metric
        if name == "mr":
            score = ranking.float().mean()
        elif name == "mrr":
            score = (1 / ranking.float()).mean()
        elif name.startswith("hits@"):
            values = name[5:].split("_")
            threshold = int(values[0])
            if len(values) > 1:
                num_sample = int(values[1])
                fp_rate = (ranking - 1).float() / num_negative      # unbiased estimation, run.py:208-217
                score = 0
                for i in range(threshold):
                    num_comb = math.factorial(num_sample - 1) / math.factorial(i) / math.factorial(num_sample - i - 1)
                    score = score + num_comb * (fp_rate ** i) * ((1 - fp_rate) ** (num_sample - i - 1))
                score = score.mean()
            else:
                score = (ranking <= threshold).float().mean()
        else:
            raise ValueError("Unknown metric `%s`" % metric)
        out[metric] = float(score)
    return out


@torch.no_grad()
def evaluate(model, test_data, batch_size=8, filtered_data=None, metrics=("mr", "mrr", "hits@1", "hits@3", "hits@10"),
             max_triples=None, use_graph=True, in_flight=None, cache_relations=None):
    """Returns {metric: value} on every rank (the reference only fills it on rank 0).
    in_flight: captured evaluation steps replayed round-robin on as many streams (1 or 2; None: 2 for shards of 128 full
    batches or more, where the second capture pays for itself).
    cache_relations: compute the relation model's output for every relation once (Ultra.cache_relation_representations: it
    depends on the query relation only) instead of once per batch and direction; same bits, same metrics.  None: when the
    shard has at least as many batches as the graph has relation chunks (then the table costs less than it saves)."""
    world, rank = udist.world_size(), udist.rank()
    triples = torch.cat([test_data.target_edge_index, test_data.target_edge_type.unsqueeze(0)]).t()
    if max_triples is not None:
        triples = triples[:max_triples]
    lo, hi = udist.shard_range(len(triples), rank, world)
    mine = triples[lo:hi]
    if filtered_data is None:       # (a dataset read from triple files carries its filtering graph: data.load_triples_dir)
        filtered_data = getattr(test_data, "filtered_data", None)
    filt = test_data if filtered_data is None else filtered_data

    was_training = model.training
    model.eval()
    made_table = False
    if hasattr(model, "cache_relation_representations") and mine.is_cuda and getattr(test_data, "relation_graph", None) is not None \
            and getattr(model, "_rel_table", None) is None:
        num_rel = int(test_data.relation_graph.num_nodes)
        if cache_relations is None:
            cache_relations = 2 * len(mine) // max(batch_size, 1) >= (num_rel + batch_size - 1) // batch_size
            # (by default only where the (num_rel, num_rel, 64) fp32 table is small next to what is free: it grows with the SQUARE of
            # the relation count -- 2.6 GB at 3,200 relations -- and a second captured step may hold its activations beside it)
            if cache_relations and mine.is_cuda:
                free_bytes = torch.cuda.mem_get_info(mine.device)[0]
                cache_relations = num_rel * num_rel * 256 <= free_bytes // 8
        if cache_relations:
            model.cache_relation_representations(test_data, chunk=batch_size)
            made_table = True
    try:
        local = _local_rows(model, test_data, filt, mine, batch_size, use_graph, in_flight, triples.device)
    finally:
        if made_table:
            model.drop_relation_cache()      # (also when an exception escapes: the table must not stay on the model)
        model.train(was_training)
    flat = udist.all_gather_shards(local, len(triples), rows_per_item=2)     # the single collective of the evaluation

    ranking, num_neg, is_tail = flat[:, 0], flat[:, 1], flat[:, 2].bool()
    plain = [m for m in metrics if "-tail" not in m]
    tail = [m for m in metrics if "-tail" in m]
    out = metrics_from_rankings(ranking, num_neg, plain)
    for m in tail:
        base, direction = m.split("-")
        if direction != "tail":
            raise ValueError("Only tail metric is supported in this mode")
        out[m] = metrics_from_rankings(ranking[is_tail], num_neg[is_tail], [base])[base]
    out["_num_rankings"] = int(ranking.numel())
    return out


def _local_rows(model, test_data, filt, mine, batch_size, use_graph, in_flight, device):
    """(rank, #negatives, is_tail) rows of this rank's shard `mine`, tail and head direction, in shard order."""
    rows = []
    n_full = (len(mine) // batch_size) * batch_size
    start = 0
    if use_graph and mine.is_cuda and n_full >= 4 * batch_size:
        # Full batches: the known true answers of the whole shard are listed ONCE (one sort / unique instead of one per batch
        # and direction), then every batch is one replay of the captured step (graph.GraphedEvalStep: candidates, both
        # forwards, both rank kernels) -- the host only feeds (bs, 3) triples and two offset vectors.
        from .graph import GraphedEvalStep
        try:
            t_ptr, t_index = tasks.known_answers(filt, mine[:n_full], "tail")
            h_ptr, h_index = tasks.known_answers(filt, mine[:n_full], "head")
            # Two captured steps take the batches alternately on two streams (batches are independent; the launches of one
            # that leave the chip idle -- relation model, glue, rank kernels -- run beside the entity layers of the other;
            # graph.PipelinedForward is the same idea for the bare forward).  Reference-order plans only: the
            # re-associating plans own scratch buffers that concurrent steps would share.
            from . import rspmm
            # (a second capture costs about what it saves on a hundred batches: 130 vs 125 M scores/s on 64 batches, 170 vs 160 on 512)
            if in_flight is None:
                in_flight = 1 if os.environ.get("ULTRA_EVAL_IN_FLIGHT", "2") == "1" or n_full < 128 * batch_size else 2
            # (two steps in flight share the chip like graph.PipelinedForward: the aggregation launches of each capture on three
            # quarters of the CUs, where a layer's activations fit the last-level cache)
            from .graph import shared_launch_grid
            want_two = int(in_flight) >= 2 and n_full >= 2 * batch_size and rspmm._plan_defaults["exact_order"]
            share = want_two and 2 * batch_size * int(test_data.num_nodes) * 256 <= 128 << 20
            scope = (lambda: rspmm.tuning_scope(grid=shared_launch_grid(mine.device))) if share else (lambda: rspmm.tuning_scope())
            with scope():
                steps = [GraphedEvalStep(model, test_data, batch_size, t_index, h_index)]
                # ... if the plans the captured step really uses are all of that kind (a max-aggregate model sends its relation
                # graph to a re-associating plan), and if a second capture fits: it doubles the captured activation memory
                if want_two and all(p.exact for p in steps[0]._pinned):
                    try:
                        steps.append(GraphedEvalStep(model, test_data, batch_size, t_index, h_index))
                    except torch.cuda.OutOfMemoryError:      # one step at a time then (any other error is a real one: raised)
                        torch.cuda.synchronize()
            if share and len(steps) == 1:       # (alone after all: it gets the whole chip)
                steps = [GraphedEvalStep(model, test_data, batch_size, t_index, h_index)]
            n_slot = len(steps)
            cur = torch.cuda.current_stream(mine.device)
            streams = [cur]
            if n_slot > 1:
                # (which pair of streams interleaves is decided by a short trial: graph.pick_slot_streams)
                from .graph import pick_slot_streams

                def trial(cand, reps=6):
                    import time
                    torch.cuda.synchronize(mine.device)
                    t0 = time.perf_counter()
                    for i in range(reps):
                        with torch.cuda.stream(cand[i % n_slot]):
                            steps[i % n_slot](mine[:batch_size], t_ptr[:batch_size + 1], h_ptr[:batch_size + 1])
                    torch.cuda.synchronize(mine.device)
                    return (time.perf_counter() - t0) / reps
                streams, _report = pick_slot_streams(mine.device, n_slot, trial)
            out = torch.empty(n_full // batch_size, 2 * batch_size, 3, dtype=torch.long, device=mine.device)
            for s in streams:
                if s is not cur:
                    s.wait_stream(cur)          # the known-answer lists, `mine`, `out` were produced on the caller's stream
            for b in range(n_full // batch_size):
                lo = b * batch_size
                k = b % n_slot
                with torch.cuda.stream(streams[k]):
                    out[b].copy_(steps[k](mine[lo:lo + batch_size], t_ptr[lo:lo + batch_size + 1], h_ptr[lo:lo + batch_size + 1]),
                                 non_blocking=True)
            for s in streams:
                if s is not cur:
                    cur.wait_stream(s)
            rows.append(out.view(-1, 3))
            start = n_full
            for s in streams:
                s.synchronize()                 # (the captures' buffers go back to the allocator below: nothing may still run in them)
            del steps
        except models.NotOnFusedPath:       # model outside the fused inference path: everything runs eagerly below
            torch.cuda.synchronize()
    for start in range(start, len(mine), batch_size):
        batch = mine[start:start + batch_size]
        t_batch, h_batch = tasks.all_negative(test_data, batch)
        t_pred = model(test_data, t_batch)
        h_pred = model(test_data, h_batch)
        # filtered rank of the positives and their number of negatives (tasks.py:94-141): one fused kernel per direction
        # on the GPU (no (bs, N) masks); the mask-based formulation of the reference with the same interface elsewhere
        rank_fn = tasks.filtered_ranking if t_pred.is_cuda else tasks.filtered_ranking_masks
        t_rank, t_neg = rank_fn(filt, batch, t_pred, mode="tail")
        h_rank, h_neg = rank_fn(filt, batch, h_pred, mode="head")
        is_tail = torch.ones_like(t_rank)
        rows.append(torch.stack([t_rank, t_neg, is_tail], dim=-1))
        rows.append(torch.stack([h_rank, h_neg, torch.zeros_like(is_tail)], dim=-1))
    if rows:
        return torch.cat(rows).long()
    return torch.zeros(0, 3, dtype=torch.long, device=device)
