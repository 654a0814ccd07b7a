"""`bench.py --stub-cpu`: the N-rank program of the benchmark without GPUs.  No multi-GPU box has ever been
available to this repository, so the 8-rank control flow -- self-launch under torch.distributed.run, process group, rank 0's
readout order broadcast to the others, the per-step all-gather behind each of the two pipeline slots, barrier-fenced
max-over-ranks timing, per-rank digests, ONE JSON line -- is exercised over gloo on CPU around a stub scorer.  The line labels
itself as a dry run; nothing here is a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-cpu"] + extra, capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, **(env or {})))
    return r


@pytest.mark.parametrize("world", [8, 2])
def test_stub_ranks_print_one_line_of_the_contract(world):
    r = _run(["--gpus", str(world), "--steps", "5", "--warmup", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "rank 0 prints exactly one line: %r" % lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == world and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert "stub" in d["data"] and "NOT a measurement" in d["data"]
    pr = d["config"]["per_rank"]
    assert len(pr["ms_per_step"]) == world and len(pr["readout_order_id"]) == world
    assert pr["probe_scores_identical"] and pr["readout_order_identical"]
    # every step's all-gather returned world x batch score rows, rank r's block = what rank r scored for its own batch
    assert pr["gathered_rows_per_step"] == world * d["config"]["batch_per_gpu"] and pr["gathered_blocks_in_rank_order"]
    # whole-job aggregate over the MAX of the ranks' clocks
    assert d["ms_per_step"] == pytest.approx(max(pr["ms_per_step"]), rel=1e-9)
    assert d["value"] == pytest.approx(world * d["config"]["triples_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3), rel=1e-9)


def test_stub_refuses_a_world_size_that_contradicts_gpus():
    r = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "does not match" in (r.stderr + r.stdout)
