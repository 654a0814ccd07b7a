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


RELATION_TABLE_MAX_BYTES = 1 << 30      # evaluate() builds the relation table by default only below this size
MAX_IN_FLIGHT = 3                       # captured evaluation steps replayed round-robin (bench.py found three worth 3 - 4 % over two)
# the streams the captured steps run on, chosen once per process and device by a short trial on REAL batches (graph.pick_slot_streams)
_SLOT_STREAMS = {}


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
             max_triples=None, use_graph=True, in_flight=None, cache_relations=None, stats=None):
    """Returns {metric: value} on every rank (the reference only fills it on rank 0).
    in_flight: captured evaluation steps replayed round-robin on as many streams (1 to 3; None: 3 for shards of 256 full
    batches or more, 2 from 128, where the extra captures pay for themselves).
    cache_relations: compute the relation model's output for every relation once (Ultra.cache_relation_representations: it
    depends on the query relation only) instead of once per batch and direction; same bits, same metrics.  None: when the
    shard has at least as many batches as the graph has relation chunks (then the table costs less than it saves).
    stats: a dict that receives where the time went on this rank (seconds: "table", "capture", "trial", "replay"; "in_flight",
    "batches", "slot_streams") -- measurements; the timers synchronise the device, so leave it None otherwise."""
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
            # decided from quantities every rank agrees on (the shard size of the smallest rank, the relation count) -- never from
            # a rank's own free memory, or ranks could take different paths -- and only where the
            # (num_rel, num_rel, 64) fp32 table is small: it grows with the SQUARE of the relation count, 2.6 GB at 3,200 relations
            per_rank = len(triples) // max(world, 1)
            cache_relations = (2 * per_rank // max(batch_size, 1) >= (num_rel + batch_size - 1) // batch_size
                               and num_rel * num_rel * 256 <= RELATION_TABLE_MAX_BYTES)
        if cache_relations:
            t0 = _tick(stats, mine.device)
            model.cache_relation_representations(test_data, chunk=batch_size)
            made_table = True
            _tock(stats, "table", t0, mine.device)
    try:
        local = _local_rows(model, test_data, filt, mine, batch_size, use_graph, in_flight, triples.device, stats)
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


def _tick(stats, device):
    if stats is None:
        return None
    import time
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return time.perf_counter()


def _tock(stats, key, t0, device):
    if stats is not None:
        stats[key] = stats.get(key, 0.0) + _tick(stats, device) - t0


def _local_rows(model, test_data, filt, mine, batch_size, use_graph, in_flight, device, stats=None):
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
            n_batch = n_full // batch_size
            if in_flight is None:
                env = os.environ.get("ULTRA_EVAL_IN_FLIGHT")
                in_flight = int(env) if env else (1 if n_batch < 128 else (2 if n_batch < 256 else 3))
            in_flight = max(1, min(int(in_flight), MAX_IN_FLIGHT, n_batch))
            # (several steps in flight share the chip like graph.PipelinedForward: the aggregation launches of each capture on three
            # quarters of the CUs, where a layer's activations fit the last-level cache)
            from .graph import shared_launch_grid
            want_more = in_flight >= 2 and rspmm._plan_defaults["exact_order"]
            share = want_more and 2 * batch_size * int(test_data.num_nodes) * 256 <= 128 << 20
            scope = (lambda: rspmm.tuning_scope(grid=shared_launch_grid(mine.device))) if share else (lambda: rspmm.tuning_scope())
            t_cap = _tick(stats, mine.device)
            with scope():
                steps = [GraphedEvalStep(model, test_data, batch_size, t_index, h_index)]
                # ... if the plans the captured step really uses are all of that kind (a max-aggregate model sends its relation
                # graph to a re-associating plan), and while another capture fits: each holds its own activation memory
                while want_more and len(steps) < in_flight and all(p.exact for p in steps[0]._pinned):
                    try:
                        steps.append(GraphedEvalStep(model, test_data, batch_size, t_index, h_index))
                    except (torch.cuda.OutOfMemoryError, RuntimeError) as err:
                        # fewer steps at a time then; on ROCm a capture that runs out of memory may surface as a plain
                        # RuntimeError (hipErrorOutOfMemory from hipGraphInstantiate): anything else is a real error
                        if not isinstance(err, torch.cuda.OutOfMemoryError) and "out of memory" not in str(err).lower():
                            raise
                        torch.cuda.synchronize()
                        break
            if share and len(steps) == 1:       # (alone after all: it gets the whole chip)
                steps = [GraphedEvalStep(model, test_data, batch_size, t_index, h_index)]
            n_slot = len(steps)
            _tock(stats, "capture", t_cap, mine.device)
            if stats is not None:
                stats.update(in_flight=n_slot, batches=n_batch)
            cur = torch.cuda.current_stream(mine.device)
            out = torch.empty(n_batch, 2 * batch_size, 3, dtype=torch.long, device=mine.device)

            def run(streams, b_lo, b_hi):
                """Enqueue batches [b_lo, b_hi): step k always on streams[k] (a step's buffers belong to one stream at a time)."""
                for b in range(b_lo, b_hi):
                    lo = b * batch_size
                    k = b % n_slot
                    with torch.cuda.stream(streams[k]):
                        out[b].copy_(steps[k](mine[lo:lo + batch_size], t_ptr[lo:lo + batch_size + 1], h_ptr[lo:lo + batch_size + 1]),
                                     non_blocking=True)

            streams = [cur]
            done = 0
            if n_slot > 1:
                # Which set of streams interleaves well is decided by a short trial (graph.pick_slot_streams) -- on REAL batches,
                # whose rows are kept, so the trial discards no work (216 replays of a dummy batch per evaluate() call
                # cost more than the second capture saved on shards of a few hundred batches) -- once per process and device;
                # shards too small to spare the trial's batches take the first candidate.
                key = (str(mine.device), n_slot)
                streams = _SLOT_STREAMS.get(key)
                if streams is None:
                    from .graph import pick_slot_streams, slot_stream
                    reps = 2 * n_slot
                    n_cand = min(int(os.environ.get("ULTRA_SLOT_STREAM_CANDIDATES", "12")), n_batch // (6 * reps))
                    if n_cand >= 2:
                        cursor = [0]

                        def trial(cand):
                            import time
                            cur.synchronize()
                            for s in cand:
                                s.wait_stream(cur)
                            t0 = time.perf_counter()
                            run(cand, cursor[0], cursor[0] + reps)
                            for s in cand:
                                s.synchronize()
                            cursor[0] += reps
                            return (time.perf_counter() - t0) / reps
                        t_trial = _tick(stats, mine.device)
                        try:
                            streams, _report = pick_slot_streams(mine.device, n_slot, trial, n_cand=n_cand)
                        finally:
                            done = cursor[0]
                        _tock(stats, "trial", t_trial, mine.device)
                        if stats is not None:
                            stats.update(slot_streams=_report, trial_batches=done)
                    else:
                        streams = [slot_stream(mine.device) for _ in range(n_slot)]
                    _SLOT_STREAMS[key] = streams
            for s in streams:
                if s is not cur:
                    s.wait_stream(cur)          # the known-answer lists, `mine`, `out` were produced on the caller's stream
            t_run = _tick(stats, mine.device)
            run(streams, done, n_batch)
            for s in streams:
                if s is not cur:
                    cur.wait_stream(s)
            _tock(stats, "replay", t_run, mine.device)
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
