"""Where a fine-tuning step's wall time goes (FB15k237 shape by default): per phase -- negative sampling, forward, loss,
backward, AdamW -- (a) the time with a device synchronise after each phase (GPU time of the phase, launch gaps included),
(b) the HOST time to issue the phase with nothing synchronised (a phase whose issue time is near its GPU time is
launch-bound), and the step as the benchmark times it.  One JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import secondary_bench as sb  # noqa: E402
from ultra_amd import synthetic, tasks  # noqa: E402

dev = sb.dev


def main(shape="fb15k237", bs=8, num_negative=256, aggr="sum", iters=10):
    data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234).to(dev)
    model = sb.load_model(aggr, "ultra_50g").train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4)
    triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)[: data.num_edges // 2]
    names = ("sampling", "forward", "loss", "backward", "adamw")

    def phases(i):
        batch = triples[(i * bs) % 4096:(i * bs) % 4096 + bs]
        neg = tasks.negative_sampling(data, batch, num_negative, strict=True)
        yield
        pred = model(data, neg)
        yield
        target = torch.zeros_like(pred)
        target[:, 0] = 1
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target, reduction="none")
        neg_w = torch.ones_like(pred)
        with torch.no_grad():
            neg_w[:, 1:] = torch.softmax(pred[:, 1:], dim=-1)
        loss = ((loss * neg_w).sum(dim=-1) / neg_w.sum(dim=-1)).mean()
        yield
        opt.zero_grad()
        loss.backward()
        yield
        opt.step()
        yield

    def run(i, sync):
        marks = []
        t = time.perf_counter()
        for _ in phases(i):
            if sync:
                torch.cuda.synchronize()
            now = time.perf_counter()
            marks.append(now - t)
            t = now
        return marks

    for i in range(3):
        run(i, False)
    torch.cuda.synchronize()
    synced = [0.0] * 5
    for i in range(iters):
        for k, v in enumerate(run(3 + i, True)):
            synced[k] += v / iters
    issue = [0.0] * 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        for k, v in enumerate(run(20 + i, False)):
            issue[k] += v / iters
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / iters
    print(json.dumps({"shape": shape, "aggregate": aggr, "ms_per_step": 1e3 * step,
                      "gpu_ms_by_phase_synchronised": {n: round(1e3 * v, 3) for n, v in zip(names, synced)},
                      "host_issue_ms_by_phase": {n: round(1e3 * v, 3) for n, v in zip(names, issue)},
                      "sum_synchronised": round(1e3 * sum(synced), 3), "sum_issue": round(1e3 * sum(issue), 3)}), flush=True)


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["fb15k237"]), aggr=(sys.argv[2] if len(sys.argv) > 2 else "sum"))
