"""A long run of the captured fine-tuning step against the same steps launched kernel by kernel: N AdamW steps each from one
initialisation and one seed (same negatives), then the parameters of the two models compared BIT FOR BIT and the loss curves side by
side.  tests/test_train_gpu.py does this for six steps on a small graph; this is the FB15k237-shaped graph and hundreds of steps --
any rare ordering slip in a hand-off (LDS flags of the update backward, the sampler's stream) would show as a differing bit.
Usage: python tools/train_soak_probe.py [steps] [shape]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import secondary_bench as sb  # noqa: E402
from ultra_amd import synthetic, tasks, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
shape = sys.argv[2] if len(sys.argv) > 2 else "fb15k237"
bs, num_negative = 8, 256
data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234).to(sb.dev)
triples = torch.stack([data.edge_index[0], data.edge_index[1], data.edge_type], dim=-1)[: data.num_edges // 2]


def positives():
    i = 0
    while True:
        yield triples[(i * bs) % 4096:(i * bs) % 4096 + bs]
        i += 1


def run(captured):
    torch.manual_seed(7)
    model = sb.load_model("sum", "ultra_50g").train()
    opt = train.make_adamw(model, capturable=captured)
    negatives = tasks.prefetch_negatives(positives(), data, num_negative, strict=True)
    losses = []
    if captured:
        first = next(negatives)
        step = train.GraphedTrainStep(model, data, opt, first, num_negative=num_negative)
        losses.append(step(first).clone())
        for _ in range(steps - 1):
            losses.append(step(next(negatives)).clone())
        step.check()
    else:
        for _ in range(steps):
            losses.append(train.train_step(model, data, next(negatives), opt, num_negative=num_negative).detach().clone())
    torch.cuda.synchronize()
    return [p.detach().clone() for p in model.parameters()], torch.stack([l.reshape(()) for l in losses]).cpu()


pe, le = run(False)
pc, lc = run(True)
pc2, lc2 = run(True)
same = sum(int(torch.equal(a, b)) for a, b in zip(pe, pc))
same2 = sum(int(torch.equal(a, b)) for a, b in zip(pc, pc2))
print("%s, %d steps: parameters bit-equal eager vs captured: %d of %d tensors; captured vs captured again: %d of %d; losses equal at %d of %d steps "
      "(first %.6f, last %.6f eager / %.6f captured); max |d parameter| %.3g" % (
          shape, steps, same, len(pe), same2, len(pe), int((le == lc).sum()), steps, le[0], le[-1], lc[-1],
          max(float((a - b).abs().max()) for a, b in zip(pe, pc))), flush=True)
