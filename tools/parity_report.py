"""Where do the last 1e-4 of score difference come from?  Runs on the GPU box.

Compares, on the benchmark graph and batch: the HIP engine (fp32, fast plan and exact-order plan), the CPU
oracle in fp32 (= the reference's arithmetic) and the CPU oracle in fp64 (the 'truth' both approximate)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ultra_oracle_model  # noqa: E402
from ultra_amd import models, rspmm, synthetic, tasks  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="fb15k237")
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    dev = torch.device("cuda:0")
    data = synthetic.make_kg(**synthetic.SHAPES[args.shape], seed=1234)
    cfg = synthetic.default_model_cfg()
    state = torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                    "ultra_3g_model.pt"))
    state64 = {k: v.double() for k, v in state.items()}
    model = models.Ultra(**cfg)
    model.load_state_dict(state)
    model = model.to(dev).eval()
    gdata = data.to(dev)
    fn = ultra_oracle_model.reference_rspmm_fn()
    rep = []
    for b in range(args.batches):
        batch = data.target_triples[b * args.bs:(b + 1) * args.bs]
        t_batch, _ = tasks.all_negative(data, batch)
        cpu32 = ultra_oracle_model.ultra_forward(state, cfg, data, t_batch, rspmm_fn=fn)
        cpu64 = ultra_oracle_model.ultra_forward(state64, cfg, data, t_batch)
        with torch.no_grad():
            rspmm.set_plan_defaults()
            gpu_fast = model(gdata, t_batch.to(dev)).cpu()
            rspmm.set_plan_defaults(exact_order=True)
            gpu_exact = model(gdata, t_batch.to(dev)).cpu()
            rspmm.set_plan_defaults()
        t_mask, _ = tasks.strict_negative_mask(data, batch)
        pos_t = batch[:, 1]
        rk = {k: tasks.compute_ranking(v, pos_t, t_mask) for k, v in
              dict(cpu32=cpu32, cpu64=cpu64.float(), gpu_fast=gpu_fast, gpu_exact=gpu_exact).items()}
        rec = dict(batch=b, score_abs_max=cpu64.abs().max().item(), score_std=cpu64.std().item(),
                   err_cpu32_vs_64=(cpu32.double() - cpu64).abs().max().item(),
                   err_gpu_fast_vs_64=(gpu_fast.double() - cpu64).abs().max().item(),
                   err_gpu_exact_vs_64=(gpu_exact.double() - cpu64).abs().max().item(),
                   diff_gpu_fast_vs_cpu32=(gpu_fast - cpu32).abs().max().item(),
                   diff_gpu_exact_vs_cpu32=(gpu_exact - cpu32).abs().max().item(),
                   rank_mismatch_gpu_fast_vs_cpu32=int((rk["gpu_fast"] != rk["cpu32"]).sum()),
                   rank_mismatch_gpu_exact_vs_cpu32=int((rk["gpu_exact"] != rk["cpu32"]).sum()),
                   rank_mismatch_cpu32_vs_cpu64=int((rk["cpu32"] != rk["cpu64"]).sum()),
                   rank_mismatch_gpu_fast_vs_cpu64=int((rk["gpu_fast"] != rk["cpu64"]).sum()),
                   ranks_cpu32=rk["cpu32"].tolist(), ranks_gpu=rk["gpu_fast"].tolist(), ranks_cpu64=rk["cpu64"].tolist())
        rep.append(rec)
        print(json.dumps(rec), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/parity_report.json", "w"), indent=1)


if __name__ == "__main__":
    main()
