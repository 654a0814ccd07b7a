"""Fast inference path vs the generic path (every A/B switch off) at the benchmark shape for several batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ultra_amd import layers, models, synthetic, tasks

dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"]).to(dev)
model = models.Ultra(**synthetic.default_model_cfg())
model.load_state_dict(torch.load(os.path.join(ROOT, "tests", "golden", "ultra_3g_model.pt")))
model = model.to(dev).eval()
worst = 0.0
for bs in (1, 2, 3, 5, 8, 13, 16, 33, 64):
    t_batch, h_batch = tasks.all_negative(data, data.target_triples[100:100 + bs])
    with torch.no_grad():
        fast = [model(data, b).clone() for b in (t_batch, h_batch)]
        layers.ONEHOT_FAST_PATH = layers.POINT_BOUNDARY_FAST_PATH = layers.FUSED_DENSE_LAYER = False
        models.PROLOGUE_FAST_PATH = False
        try:
            slow = [model(data, b).clone() for b in (t_batch, h_batch)]
        finally:
            layers.ONEHOT_FAST_PATH = layers.POINT_BOUNDARY_FAST_PATH = layers.FUSED_DENSE_LAYER = True
            models.PROLOGUE_FAST_PATH = True
    err = max((a - b).abs().max().item() for a, b in zip(fast, slow))
    worst = max(worst, err)
    print("bs %3d  max |fast - generic| = %.2e   finite=%s" % (bs, err, all(torch.isfinite(a).all().item() for a in fast)), flush=True)
assert worst <= 5e-5, worst
print("ok")
