"""Why is the captured fine-tuning step slower inside bench.py's process (3.2 - 3.6 ms) than in a fresh one (2.9)?  The same
secondary_bench.train_case after (a) nothing, (b) N idle streams of both priorities, (c) a PipelinedForward as bench.py builds it
(stream trial, three captures, 64 pre-warm steps) that is deleted again.  One process per variant: PROBE_VARIANT=a|b|c."""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import secondary_bench  # noqa: E402
from ultra_amd import rspmm, synthetic, tasks  # noqa: E402

variant = os.environ.get("PROBE_VARIANT", "a")
dev = torch.device("cuda:0")
keep = []
if variant == "b":
    for i in range(int(os.environ.get("PROBE_STREAMS", "40"))):
        s = torch.cuda.Stream(priority=-1 if i % 2 else 0)
        with torch.cuda.stream(s):
            torch.zeros(8, device=dev).add_(1)
        keep.append(s)
    torch.cuda.synchronize()
elif variant in ("c", "d"):
    from ultra_amd.graph import PipelinedForward
    data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(dev)
    model = secondary_bench.load_model("sum", "ultra_3g").eval()
    batch = tasks.all_negative(data, data.target_triples[:8])[0]
    piped = PipelinedForward(model, data, batch, depth=3)
    for _ in range(64):
        piped(batch)
    piped.join()
    torch.cuda.synchronize()
    if variant == "c":
        del piped
    del model, data
    gc.collect()
    rspmm.clear_plan_cache()
    torch.cuda.empty_cache()
out = secondary_bench.train_case("fb15k237")
print(variant, "ms_per_step %.3f  eager %.3f" % (out["ms_per_step"], out["ms_per_step_eager"]), flush=True)
