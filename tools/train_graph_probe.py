"""The fine-tuning step launched one kernel at a time against the same step as ONE hipGraph replay (ultra_amd/train.py):
ms per step for both (negatives one batch ahead on the sampler's side stream in both), and the losses of the first steps side
by side.  One JSON line per shape.  Usage: python tools/train_graph_probe.py [fb15k237] [yago310] ..."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import secondary_bench as sb  # noqa: E402
from ultra_amd import synthetic, tasks, train  # noqa: E402

dev = sb.dev


def run(shape, bs=8, num_negative=256, aggr="sum", steps=20):
    data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234).to(dev)
    triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)[: data.num_edges // 2]

    def positives():
        i = 0
        while True:
            yield triples[(i * bs) % 4096:(i * bs) % 4096 + bs]
            i += 1

    out = {"shape": shape, "batch": bs, "num_negative": num_negative, "aggregate": aggr}
    only = os.environ.get("PROBE_ONLY", "")
    if only != "captured":
        out.update(eager(data, positives, num_negative, aggr, steps))
    if only != "eager":
        out.update(captured(data, positives, num_negative, aggr, steps))
    return out


def eager(data, positives, num_negative, aggr, steps):
    out = {}
    torch.manual_seed(7)
    model = sb.load_model(aggr, "ultra_50g").train()
    opt = train.make_adamw(model)
    negatives = tasks.prefetch_negatives(positives(), data, num_negative, strict=True)
    losses = []

    def eager_step():
        losses.append(train.train_step(model, data, next(negatives), opt, num_negative=num_negative))
    out["eager_ms"] = 1e3 * sb.timeit(eager_step, 3, steps)
    out["eager_loss"] = [round(l.item(), 6) for l in losses[:6]]
    import time
    torch.cuda.synchronize()
    issue = 0.0
    for _ in range(steps):
        t = time.perf_counter()
        eager_step()
        issue += time.perf_counter() - t
    torch.cuda.synchronize()
    out["eager_host_issue_ms"] = 1e3 * issue / steps
    return out


def captured(data, positives, num_negative, aggr, steps):
    out = {}
    torch.manual_seed(7)
    model = sb.load_model(aggr, "ultra_50g").train()
    opt = train.make_adamw(model, capturable=True)
    negatives = tasks.prefetch_negatives(positives(), data, num_negative, strict=True)
    first = next(negatives)
    step = train.GraphedTrainStep(model, data, opt, first, num_negative=num_negative)
    losses = [step(first).clone()]

    def graph_step():
        losses.append(step(next(negatives)).clone())
    out["captured_ms"] = 1e3 * sb.timeit(graph_step, 2, steps)
    out["captured_loss"] = [round(l.item(), 6) for l in losses[:6]]
    # host time to ISSUE a step (sampler's launches for the next batch + input copy + replay; nothing synchronised but the
    # sampler's own stream) against the step's time: who sets the pace
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    issue = 0.0
    for _ in range(steps):
        t = time.perf_counter()
        step(next(negatives))
        issue += time.perf_counter() - t
    torch.cuda.synchronize()
    out["captured_host_issue_ms"] = 1e3 * issue / steps
    out["captured_ms_second_loop"] = 1e3 * (time.perf_counter() - t0) / steps
    step.check()
    return out


if __name__ == "__main__":
    for shape in (sys.argv[1:] or ["fb15k237", "yago310"]):
        print(json.dumps(run(shape)), flush=True)
