"""Headline benchmark: triples scored/sec, all-tail ranking on an FB15k237-shaped graph (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one Ultra.forward(data, t_batch) per GPU with t_batch = (bs=8, N, 3) all-tail candidates
(script/run.py:135-136): 12 relational SpMM calls + the dense layer updates + the readout MLP, scoring
bs * N triples -- with the plans that sum in the REFERENCE'S ORDER (rspmm.cpp:61-72; ultra_amd's default).  Queries
shard over ranks (each rank scores its own 8 queries, graph + weights replicated); with N > 1 GPUs every step ends
with one RCCL all-gather of the per-rank score rows.  Weak scaling: per-GPU work is fixed.  Rank 0 prints ONE JSON line.
The forward is a captured hipGraph; by default THREE captures take the steps alternately on as many streams (--in-flight 3,
ultra_amd/graph.py PipelinedForward): steps are independent batches, and the launches of one that leave the chip idle run
beside the entity layers of the next.  ms_per_step = elapsed / steps; `modes.one_batch_in_flight` has the one-stream figure.

Extra blocks of the JSON line (rank 0, N = 1):
  roofline     -- the dominant kernel (one entity layer: reference-order rspmm + the layer update in its tail,
                  ultra_rspmm_forward_update) at the benchmark point AND at the
                  HBM-bound point (CoDEx-L shape, batch 8: x + out = 319 MB > the 256 MB Infinity Cache): kernel time
                  from HIP events on the launch stream, HBM-side and L2 bytes from rocprofv3 --pmc passes run by this
                  script (FETCH_SIZE / WRITE_SIZE / TCC_REQ in separate passes, calibrated on a 1 GiB stream copy in the
                  same pass), algorithmic byte models beside them.  `achieved` = measured HBM-side GB/s.
  cpu_baseline -- the oracle port of Ultra.forward (torch CPU ops + the reference's own rspmm.cpp TU when oracle/_ref is
                  present) on the host cores, same workload, bounded sample.
  parity       -- GPU scores / rankings against that CPU result on the identical batch.
"""
import os as _os

# Hardware queues.  The HIP runtime maps a process's streams round-robin onto GPU_MAX_HW_QUEUES hardware queues (default 4) per
# priority level, and a launcher's rank (RCCL communicator = more streams) used to land its two pipeline slots' streams on queues
# that interleave worse than the plain process's: round 4 pinned GPU_MAX_HW_QUEUES=3 for launched ranks (0.605 - 0.625 -> 0.583
# ms per step).  Since round 5 the pipeline picks its slots' streams by measurement (ultra_amd/graph.py pick_slot_streams: a
# normal- and a high-priority pair timed for a few steps when the pipeline is built; the launcher's rank ends up on the
# high-priority pair, the plain process on the normal one), and the override is gone -- the runtime's default applies unless the
# caller sets the variable (ULTRA_BENCH_LAUNCHER_QUEUES=n re-creates the old override for A/B runs; the runtime reads the
# variable when it is loaded, hence here, before torch is imported).
if "RANK" in _os.environ and "WORLD_SIZE" in _os.environ and _os.environ.get("ULTRA_BENCH_LAUNCHER_QUEUES"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", _os.environ["ULTRA_BENCH_LAUNCHER_QUEUES"])

import argparse
import contextlib
import csv
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
L2_PEAK_GBS = 34500.0      # MI355X_MICROARCH.md: aggregate L2 bandwidth
COPY_BYTES = 1 << 30
ROOFLINE_POINTS = [("fb15k237", 8), ("codex_l", 8)]
# measurement switch: ULTRA_BENCH_UPDATE_FORM=1 / 2 / 3 runs every one-launch layer of this process with the update in the
# kernel's tail / beside the walk (rows by reference) / beside the walk (rows through LDS) -- ultra_tuning.reserved[2], DESIGN.md
# 3.8 -- instead of the library's choice (form 3 on graphs of 10+ steps a row, which both roofline points are); the roofline
# block then describes that kernel
UPDATE_FORM = int(os.environ.get("ULTRA_BENCH_UPDATE_FORM") or 0)
ORDER_KERNEL = "rspmm_order_kernel<float, 0, 0, true, false, true, %d>" % (UPDATE_FORM or 3)   # (..., STREAMS, UPDATE): aggregate + layer update
# the vector L1 / texture-address path of a CU delivers 64 B per clock (MI355X_MICROARCH.md: 16-B-per-lane loads, four lanes
# per clock); at the 2.4 GHz boost clock the 256 CUs gather 39.3 TB/s -- the roof of a kernel whose gathers hit in L2
L1_PEAK_GBS = 256 * 64 * 2.4


def available_cores():
    """Host cores this process may actually use: min(cpu_count, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def b_gather(E, N, R, D):
    # SURVEY.md section 8d, gather model: every edge reads its source row, output written once, relation table once,
    # CSR records once (8 B / edge here: (col, type) pairs), item list once.
    return 4 * D * (E + N + R) + 8 * E + 16 * N


def b_min(E, N, R, D):
    # compulsory model: x read once, out written once, relation table once, records once
    return 4 * D * (2 * N + R) + 8 * E + 16 * N


UPDATE_WEIGHT_BYTES = 4 * (64 * 128 + 3 * 64)      # Linear(128 -> 64) + bias + LayerNorm weight / bias


def b_gather_layer(E, N, R, D):
    # the one-launch layer (ultra_rspmm_forward_update): the rspmm above (its "output" is the aggregate), then the update reads x
    # and the aggregate back and writes the layer output
    return b_gather(E, N, R, D) + 3 * 4 * D * N + 4 * N + UPDATE_WEIGHT_BYTES


def b_min_layer(E, N, R, D):
    # compulsory for the layer: x read once, LAYER OUTPUT written once (the aggregate is an intermediate: every byte of its
    # round trip counts as avoidable traffic), relation table, records, row lists, weights once
    return b_min(E, N, R, D) + 4 * N + UPDATE_WEIGHT_BYTES


@contextlib.contextmanager
def _stdout_to_stderr():
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)      # C stdio buffers of the libraries that printed
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)


# ---------------------------------------------------------------------------------------------------------------------
# roofline measurement
# ---------------------------------------------------------------------------------------------------------------------
def _point_operands(shape, bs, dev):
    from ultra_amd import rspmm, synthetic
    if UPDATE_FORM:
        rspmm.set_tuning(update_form=UPDATE_FORM)
    data = synthetic.make_kg(**synthetic.SHAPES[shape], seed=1234, relation_graph=False)
    N, R = data.num_nodes, data.num_relations
    g = torch.Generator().manual_seed(0)
    x = torch.randn(bs, N, 64, generator=g).to(dev)
    rel = torch.randn(bs, R, 64, generator=g).to(dev)
    # the boundary condition as the forward passes it: one row per sample (ultra_rspmm_forward_point)
    point = (data.target_triples[:bs, 0].contiguous().to(dev), torch.randn(bs, 64, generator=g).to(dev))
    plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
    # the layer update's parameters (Linear(128 -> 64), LayerNorm(64)); flags 7 = LayerNorm | ReLU | residual
    upd = ((torch.randn(64, 128, generator=g) / 11).to(dev),) + tuple(torch.randn(64, generator=g).to(dev) for _ in range(3))
    return data, plan, rel, x, point, upd


def _stream_copy(dst, src):
    from ultra_amd import _lib
    _lib.check(_lib.lib.ultra_stream_copy(dst.data_ptr(), src.data_ptr(), COPY_BYTES,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


def pmc_target():
    """Child process of the --pmc passes: 1 GiB stream copy x 4, then per roofline point the entity layer (rspmm + update in
    one launch) x 4 (first of each = warm-up).  Nothing else runs on the GPU between them, in this fixed order."""
    dev = torch.device("cuda:0")
    src = torch.empty(COPY_BYTES // 4, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(4):
        _stream_copy(dst, src)
    torch.cuda.synchronize()
    del src, dst
    from ultra_amd import rspmm
    grid = int(os.environ.get("ULTRA_BENCH_LAUNCH_GRID") or 0)     # (the timed region's launch size: CUs / batches in flight)
    for shape, bs in ROOFLINE_POINTS:
        _, plan, rel, x, point, upd = _point_operands(shape, bs, dev)
        with rspmm.tuning_scope(grid=grid if shape == ROOFLINE_POINTS[0][0] else 0):   # (the second point: whole-chip launches)
            for _ in range(4):
                if plan.forward_update(rel, x, upd[0], upd[1], upd[2], upd[3], 1e-5, 7, point=point) is None:
                    raise RuntimeError("the one-launch layer does not serve this point")
        torch.cuda.synchronize()
        del plan, rel, x


def _run_pmc_pass(counters, timeout=240, launch_grid=0):
    """One rocprofv3 counter pass over pmc_target(); returns {counter: {"copy": [..], "points": [[..], [..]]}} or raises."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="ultra_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", ULTRA_BENCH_LAUNCH_GRID=str(launch_grid))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--",
                                      sys.executable, os.path.abspath(__file__), "--pmc-target"]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            raise RuntimeError("rocprofv3 pass failed (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
        keep = os.environ.get("ULTRA_BENCH_PMC_KEEP")       # raw per-dispatch counters, for profiles/
        if keep:
            os.makedirs(keep, exist_ok=True)
            shutil.copy(files[0], os.path.join(keep, "_".join(counters) + "_counter_collection.csv"))
        out = {c: {"copy": [], "kernel": []} for c in counters}
        for row in csv.DictReader(open(files[0])):
            c = row["Counter_Name"]
            if c not in out:
                continue
            name = row["Kernel_Name"]
            key = "copy" if "stream_copy_kernel" in name else ("kernel" if ORDER_KERNEL in name else None)
            if key:
                out[c][key].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        res = {}
        for c, d in out.items():
            copy = [v for _, v in sorted(d["copy"])][1:]                    # drop the warm-up launch
            ker = [v for _, v in sorted(d["kernel"])]
            per = len(ker) // len(ROOFLINE_POINTS)
            if not copy or per < 2:
                raise RuntimeError("unexpected dispatch counts in the %s pass" % c)
            res[c] = {"copy": sum(copy) / len(copy),
                      "points": [sum(ker[i * per + 1:(i + 1) * per]) / (per - 1) for i in range(len(ROOFLINE_POINTS))]}
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def trace_target(steps=40):
    """Child process of the --kernel-trace passes: the benchmark's forward replayed `steps` times -- one batch at a time (one
    captured hipGraph, one stream), or with ULTRA_BENCH_TRACE_IN_FLIGHT=n as the timed region runs it (n captures on n
    streams, their aggregation kernels on CUs / n workgroups each) -- for the per-kernel durations inside the graph."""
    from ultra_amd import models, rspmm, synthetic, tasks
    from ultra_amd.graph import GraphedForward, PipelinedForward
    if UPDATE_FORM:
        rspmm.set_tuning(update_form=UPDATE_FORM)
    dev = torch.device("cuda:0")
    data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234).to(dev)
    model = models.Ultra(**synthetic.default_model_cfg())
    golden = os.path.join(ROOT, "tests", "golden", "ultra_3g_model.pt")
    if os.path.exists(golden):
        model.load_state_dict(torch.load(golden))
    model = model.to(dev).eval()
    triples = data.target_triples
    n_flight = int(os.environ.get("ULTRA_BENCH_TRACE_IN_FLIGHT") or 1)
    example = tasks.all_negative(data, triples[:8])[0]
    fwd = PipelinedForward(model, data, example, depth=n_flight) if n_flight > 1 else GraphedForward(model, data, example)
    inputs = [tasks.all_negative(data, triples[8 * i:8 * i + 8])[0] for i in range(16)]
    for i in range(steps * n_flight):
        fwd(inputs[i % 16])
    if n_flight > 1:
        fwd.join()
    torch.cuda.synchronize()


def _run_trace_pass(timeout=240, in_flight=1):
    """rocprofv3 --kernel-trace --stats over trace_target(): {kernel name: (calls, avg us)} of the captured forward's kernels
    (no counters in this pass); the stats CSV is kept under $ULTRA_BENCH_PMC_KEEP as bench_kernel_stats_inflight<n>.csv."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="ultra_trace_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", ULTRA_BENCH_TRACE_IN_FLIGHT=str(in_flight))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "trace", "--",
           sys.executable, os.path.abspath(__file__), "--trace-target"]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if r.returncode != 0 or not files:
            raise RuntimeError("rocprofv3 trace pass failed (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
        keep = os.environ.get("ULTRA_BENCH_PMC_KEEP")
        if keep:
            os.makedirs(keep, exist_ok=True)
            shutil.copy(files[0], os.path.join(keep, "bench_kernel_stats_inflight%d.csv" % in_flight))
        rows = {}
        for row in csv.DictReader(open(files[0])):
            rows[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e3)
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_roofline(dev, use_pmc=True, in_flight=1):
    """in_flight: batches the timed region keeps in flight -- with more than one its aggregation kernels are launched with three
    quarters of the CUs as workgroups (graph.PipelinedForward, shared_launch_grid): the block describes the kernel as launched
    THERE and keeps the whole-chip launch beside it."""
    from ultra_amd import _lib, rspmm
    from ultra_amd.graph import shared_launch_grid
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    launch_grid = shared_launch_grid(dev) if in_flight > 1 else 0
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "kernel": "ultra::" + ORDER_KERNEL + " (entity layer in one launch: rspmm add_mul with point boundary, relation slice "
                     "in LDS, then Linear(128->64) + LayerNorm + ReLU + residual on the rows each workgroup aggregated)"}
    # ---- stream-copy ceiling ----
    src = torch.empty(COPY_BYTES // 4, device=dev).normal_()
    dst = torch.empty_like(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _stream_copy(dst, src)
    e0.record()
    for _ in range(10):
        _stream_copy(dst, src)
    e1.record()
    torch.cuda.synchronize()
    copy_ms = e0.elapsed_time(e1) / 10
    del src, dst
    out["copy_ceiling"] = {"bytes_read_plus_written": 2 * COPY_BYTES, "ms": copy_ms,
                           "GBps": 2 * COPY_BYTES / (copy_ms * 1e-3) / 1e9,
                           "note": "ultra_stream_copy (16 B / lane, 8 loads in flight, nontemporal), 1 GiB -> 1 GiB"}
    # ---- kernel times (HIP events on the launch stream around 30 launches queued back to back) ----
    points = []
    for shape, bs in ROOFLINE_POINTS:
        data, plan, rel, x, point, upd = _point_operands(shape, bs, dev)
        E, N, R, D = data.num_edges, data.num_nodes, data.num_relations, bs * 64
        timed = plan.forward_update(rel, x, upd[0], upd[1], upd[2], upd[3], 1e-5, 7, point=point, timed=(5, 30))
        if timed is None:
            raise RuntimeError("the one-launch layer does not serve the roofline point %s" % shape)
        # 30 launches queued back to back between two HIP events on the launch stream -- the regime the kernel runs in inside the
        # captured forward (five in a row).  timed[1] has the other regime for the record: events right around ONE launch with a
        # host synchronisation after each, i.e. every launch starts on an idle chip (form 3 of the layer -- twelve waves walking
        # while four multiply -- is the one that shows the difference: ~ 100 us against ~ 87).
        ms_whole = timed[0]
        # (the second point is not the timed workload: it stays a whole-chip launch -- PipelinedForward shares the chip only
        # where the activations fit the last-level cache)
        shared = in_flight > 1 and shape == ROOFLINE_POINTS[0][0]
        ms = ms_whole
        if shared:
            with rspmm.tuning_scope(grid=launch_grid):
                ms = plan.forward_update(rel, x, upd[0], upd[1], upd[2], upd[3], 1e-5, 7, point=point, timed=(5, 30))[0]
        plan.forward_timed(rel, x, point=point, warmup=5, iters=30)      # (the aggregate alone, for the record)
        info = plan.info()
        points.append({"shape": shape, "batch": bs, "N": N, "E": E, "R": R, "D": D, "ms_per_launch": ms,
                       "workgroups_per_launch": (launch_grid or n_cu) if shared else n_cu,
                       "ms_per_launch_alone_on_the_whole_chip": ms_whole,
                       "ms_per_launch_each_on_an_idle_chip": timed[1],
                       "ms_per_launch_aggregate_only": plan.last_main_kernel_ms,
                       "x_plus_out_MB": 2 * 4 * D * N / 1e6, "chain_rows": info["n_chain_row"],
                       "gather_model_bytes": b_gather_layer(E, N, R, D), "compulsory_bytes": b_min_layer(E, N, R, D)})
        del plan, rel, x
    # ---- HBM-side and L2 traffic: rocprofv3 counter passes over the same launches (separate passes, no trace domains
    # besides --kernel-trace), calibrated on the 1 GiB copy of the same pass ----
    pmc_note = None
    if use_pmc:
        try:
            fetch = _run_pmc_pass(["FETCH_SIZE"], launch_grid=launch_grid)["FETCH_SIZE"]
            write = _run_pmc_pass(["WRITE_SIZE"], launch_grid=launch_grid)["WRITE_SIZE"]
            tcc = _run_pmc_pass(["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"], launch_grid=launch_grid)
            f_unit, w_unit = COPY_BYTES / fetch["copy"], COPY_BYTES / write["copy"]
            req_unit = 2 * COPY_BYTES / tcc["TCC_REQ_sum"]["copy"]
            out["pmc_calibration"] = {"fetch_bytes_per_FETCH_SIZE_unit": f_unit, "write_bytes_per_WRITE_SIZE_unit": w_unit,
                                      "bytes_per_TCC_REQ": req_unit,
                                      "command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python bench.py "
                                                 "--pmc-target  (C = FETCH_SIZE | WRITE_SIZE | TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum)"}
            for i, pt in enumerate(points):
                pt["hbm_read_bytes"] = fetch["points"][i] * f_unit
                pt["hbm_write_bytes"] = write["points"][i] * w_unit
                pt["hbm_bytes"] = pt["hbm_read_bytes"] + pt["hbm_write_bytes"]
                pt["l2_request_bytes"] = tcc["TCC_REQ_sum"]["points"][i] * req_unit
                hit, miss = tcc["TCC_HIT_sum"]["points"][i], tcc["TCC_MISS_sum"]["points"][i]
                pt["l2_hit_rate"] = hit / (hit + miss) if hit + miss > 0 else None
        except Exception as exc:      # no profiler on this box / pass failed: say so, never substitute a recorded number
            pmc_note = "PMC passes unavailable in this run (%s): traffic = null" % str(exc)[:200]
    else:
        pmc_note = "PMC passes skipped (--no-pmc)"
    # ---- the same kernel INSIDE the captured forward (one batch at a time): a --kernel-trace pass over the benchmark's step ----
    in_graph = None
    if use_pmc:
        try:
            def digest(stats, what):
                hit = [(n, v) for n, v in stats.items() if ORDER_KERNEL in n]
                if not hit:
                    return None
                calls, avg_us, _ = hit[0][1]
                total = sum(v[2] for v in stats.values())
                return {"kernel_avg_us": avg_us, "calls": calls, "launches_per_forward": 5,
                        "share_of_gpu_time": hit[0][1][2] / total if total else None,
                        "command": "rocprofv3 --kernel-trace --stats -- python bench.py --trace-target  (%s)" % what,
                        "top_kernels_us": [[n.split("(")[0][-90:], c, round(a, 2)] for n, (c, a, t) in
                                           sorted(stats.items(), key=lambda kv: -kv[1][2])[:8]]}
            in_graph = digest(_run_trace_pass(), "the benchmark's forward as one hipGraph on one stream, 40 replays: whole-chip launches")
            if in_flight > 1 and in_graph is not None:
                in_graph["as_timed"] = digest(_run_trace_pass(in_flight=in_flight),
                                              "ULTRA_BENCH_TRACE_IN_FLIGHT=%d: %d captures on %d streams as in the timed region, the "
                                              "aggregation kernels on %d workgroups each; a kernel's duration there includes what "
                                              "the other stream's launches cost it" % (in_flight, in_flight, in_flight, launch_grid))
        except Exception as exc:
            in_graph = {"unavailable": str(exc)[:200]}
    for pt in points:
        t = pt["ms_per_launch"] * 1e-3
        # gathers through the CU's vector L1: every edge's 256-B source row per sample (+ the update's row reads)
        pt["l1_gather_bytes"] = 4 * pt["D"] * pt["E"] + 2 * 4 * pt["D"] * pt["N"]
        pt["l1_rate_frac"] = pt["l1_gather_bytes"] / t / 1e9 / L1_PEAK_GBS
        pt["gather_model_GBps"] = pt["gather_model_bytes"] / t / 1e9
        pt["compulsory_GBps"] = pt["compulsory_bytes"] / t / 1e9
        pt["hbm_frac_compulsory"] = pt["compulsory_GBps"] / HBM_PEAK_GBS
        if "hbm_bytes" in pt:
            pt["hbm_GBps_measured"] = pt["hbm_bytes"] / t / 1e9
            pt["hbm_frac_measured"] = pt["hbm_GBps_measured"] / HBM_PEAK_GBS
            pt["traffic_over_compulsory"] = pt["hbm_bytes"] / pt["compulsory_bytes"]
            pt["l2_GBps"] = pt["l2_request_bytes"] / t / 1e9
            pt["l2_frac"] = pt["l2_GBps"] / L2_PEAK_GBS
    head, big = points[0], points[1]
    big["bound"] = "hbm"
    big["frac"] = big.get("hbm_frac_measured")
    measured = "hbm_bytes" in head
    out.update({
        "point": "%s shape, batch %d (the benchmark's call; x + out = %.0f MB: L2 / Infinity-Cache resident)"
                 % (head["shape"], head["batch"], head["x_plus_out_MB"]),
        "ms_per_launch": head["ms_per_launch"],
        "workgroups_per_launch": head["workgroups_per_launch"],
        "ms_per_launch_alone_on_the_whole_chip": head["ms_per_launch_alone_on_the_whole_chip"],
        # (the same fractions for the launch with one workgroup per CU: what the kernel does with the whole chip -- the launch
        # of the timed region occupies only `workgroups_per_launch` of the 256 CUs, the peaks above are the whole chip's)
        "whole_chip_launch": {
            "ms_per_launch": head["ms_per_launch_alone_on_the_whole_chip"],
            "frac_compulsory": head["compulsory_bytes"] / (head["ms_per_launch_alone_on_the_whole_chip"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "l1_rate_frac": head["l1_gather_bytes"] / (head["ms_per_launch_alone_on_the_whole_chip"] * 1e-3) / 1e9 / L1_PEAK_GBS,
            "frac_of_counter_traffic": (head["hbm_bytes"] / (head["ms_per_launch_alone_on_the_whole_chip"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                        if "hbm_bytes" in head else None)},
        "achieved": head["hbm_GBps_measured"] if measured else head["compulsory_GBps"],
        "achieved_definition": (("HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE, calibrated) / duration of a launch" if measured
                                 else "compulsory-model bytes / duration of a launch (no counters in this run)")
                                + ": launched as the timed region launches it, on %d workgroups (%d CUs; the rest serve the other batch "
                                "in flight)" % (head["workgroups_per_launch"], n_cu)),
        "traffic": head.get("hbm_bytes"),
        "algorithmic_bytes_per_launch": {"gather_model": head["gather_model_bytes"], "compulsory": head["compulsory_bytes"]},
        "gather_model_GBps": head["gather_model_GBps"],
        "frac_compulsory": head["hbm_frac_compulsory"],
        "traffic_over_compulsory": head.get("traffic_over_compulsory"),
        # the roof the builder argues at THIS point (x + out cache resident): the CUs' vector-L1 gather rate, not HBM.  `bound` /
        # `achieved` / `peak` / `frac` above stay the contract's HBM quantities (counter traffic / time / 8 TB/s); at
        # `hbm_bound_point` (CoDEx-L) the memory system is the roof and its own `bound` says "hbm"
        "binding_roof": {"name": "l1-gather", "frac": head["l1_rate_frac"], "peak_GBps": L1_PEAK_GBS,
                         "achieved_GBps": head["l1_gather_bytes"] / (head["ms_per_launch"] * 1e-3) / 1e9,
                         "definition": "bytes gathered through the CUs' vector L1 per launch (E x 256 B per sample + the update's row "
                                       "reads) / kernel time / (256 CUs x 64 B/clk x 2.4 GHz)"},
        "l1_rate_frac": head["l1_rate_frac"],
        "in_graph": in_graph,
        "l2_frac": head.get("l2_frac"), "l2_hit_rate": head.get("l2_hit_rate"),
        "hbm_bound_point": big,
        "points": points,
    })
    # The top-level bound / achieved / peak / frac name the roof that BINDS at the headline point (the driver's
    # parser keeps only the top level): x + out are cache resident there, the launch is bounded by the CUs' vector-L1 gather rate.
    # The contract's HBM quantities -- counter traffic / time against 8 TB/s -- stay beside it under `hbm` (and `traffic` is still
    # the PMC byte count per launch); `hbm_bound_point` (CoDEx-L) carries `bound: hbm` for the point where the memory system binds.
    out["hbm"] = {"bound": "hbm", "achieved": out["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": out["achieved"] / HBM_PEAK_GBS,
                  "achieved_definition": out["achieved_definition"], "frac_compulsory": out["frac_compulsory"]}
    out["bound"] = "l1-gather"
    out["achieved"] = out["binding_roof"]["achieved_GBps"]
    out["peak"] = L1_PEAK_GBS
    out["frac"] = out["binding_roof"]["frac"]
    out["achieved_definition"] = out["binding_roof"]["definition"]
    out["note"] = ("the kernel is the whole entity layer: twelve waves of a workgroup aggregate, four apply the update to the rows "
                   "they hand over through LDS (compulsory bytes: x in, layer output out, relation table, records, weights).  "
                   "gather-model GB/s exceeds the HBM peak where x is cache resident (every edge re-reads a 256-B source row "
                   "from L2 / Infinity Cache, not from HBM); `hbm.frac` is COUNTER traffic / time / 8 TB/s: FETCH_SIZE counts L2 "
                   "misses, Infinity-Cache (MALL) hits INCLUDED (MI355X_MICROARCH.md), so it bounds the HBM fraction from above; "
                   "`frac_compulsory` prices only the bytes the layer must move; at the headline size the binding roof is the "
                   "CUs' vector-L1 gather rate (top level: `bound: l1-gather`; the HBM quantities under `hbm`), at CoDEx-L "
                   "(`hbm_bound_point`, `bound: hbm`) it is the memory system.")
    if pmc_note:
        out["pmc_note"] = pmc_note
    return out


# ---------------------------------------------------------------------------------------------------------------------
def ranking_metrics(rank):
    """MRR / Hits@k of a vector of ranks (script/run.py:188-213), to 6 digits."""
    r = rank.double()
    out = {"mrr": round((1 / r).mean().item(), 6)}
    for k in (1, 3, 10):
        out["hits@%d" % k] = round((r <= k).double().mean().item(), 6)
    return out


def tie_band_mismatches(ref_score, got_rank, pos, mask, band):
    """Rankings that differ from the reference's by more than its own near-ties allow.  With |gpu - reference| <= d on
    every score, `gpu_pos <= gpu_c` is certain when ref_c >= ref_pos + 2 d and impossible when ref_c < ref_pos - 2 d: the
    GPU rank must lie between the reference ranks computed with the positive's score moved by +band and -band (band = 2 d)."""
    from ultra_amd import tasks
    idx = torch.arange(len(pos))
    shifted = ref_score.clone()
    shifted[idx, pos] = ref_score[idx, pos] + band
    best = tasks.compute_ranking(shifted, pos, mask)
    shifted[idx, pos] = ref_score[idx, pos] - band
    worst = tasks.compute_ranking(shifted, pos, mask)
    return int(((got_rank < best) | (got_rank > worst)).sum())


def stub_main(args):
    """`--stub-cpu`: the program of an N-rank run with everything GPU-bound replaced by a stub -- what can be checked of the
    1 / 2 / 4 / 8-GPU half of the metric where no multi-GPU box exists.  Same control flow as main(): self-launch under
    torch.distributed.run, process group (gloo), rank 0's readout order broadcast, two pipeline slots taking the steps alternately
    with the step's all-gather behind each (graph.PipelinedForward(post=)), W warm-up + exactly K timed steps between barriers,
    MAX over ranks, per-rank digests, ONE JSON line from rank 0.  The line says what it is: data = "stub"."""
    import torch.distributed as dist
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] launching %d stub ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
        sys.exit(subprocess.call(cmd, env=dict(os.environ, OMP_NUM_THREADS="1")))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d does not match the launcher's world size %d" % (args.gpus, world))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _stdout_to_stderr():      # (gloo's connection banner is written to stdout by C code; stdout carries ONE line)
            dist.init_process_group("gloo")
            dist.barrier()
    from ultra_amd import distributed as udist
    from ultra_amd import host_order, synthetic, tasks
    from ultra_amd.graph import PipelinedForward
    if world > 1 or launched:
        udist.share_readout_order(128, device=torch.device("cpu"))
    data = synthetic.make_kg(num_node=512, num_triple=4000, num_relation_base=6, num_test=20466, seed=1234, relation_graph=False)
    N, bs, triples = data.num_nodes, args.bs, data.target_triples
    gen = torch.Generator().manual_seed(7)
    emb, relv = torch.randn(N, 16, generator=gen), torch.randn(data.num_relations, 16, generator=gen)

    def score(batch):       # (bs, N, 3) -> (bs, N): any deterministic function of the batch
        h, t, r = batch.unbind(-1)
        return (emb[h] * relv[r] * emb[t]).sum(-1)

    class Slot(object):     # a "captured forward": own output buffer, valid until the slot's next call
        def __init__(self):
            self.out = torch.zeros(bs, N)

        def __call__(self, batch):
            self.out.copy_(score(batch))
            return self.out

    def batch_for(step):
        lo = ((step * world + rank) * bs) % (triples.shape[0] - bs)
        return triples[lo:lo + bs]

    n_inputs = min(args.warmup + args.steps, 256)
    inputs = [tasks.all_negative(data, batch_for(i))[0] for i in range(n_inputs)]
    piped = PipelinedForward(None, None, inputs[0], depth=args.in_flight, slot_factory=Slot)
    gather = world > 1 or launched
    seen = []

    def one_step(step):
        out = piped(inputs[step % n_inputs], post=(lambda sc: udist.all_gather_scores(sc).clone()) if gather else None)
        seen.append(tuple(out.shape))
        return out
    for i in range(args.warmup):
        one_step(i)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = one_step(args.warmup + i)
    piped.join()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64)
    per_rank = None
    slot_report = getattr(piped, "stream_report", None) or {}      # (CPU slots have no streams to choose: None per rank)
    if gather:
        probe = score(tasks.all_negative(data, triples[:bs])[0]).double()
        order_word = float(int(host_order.order_id(host_order.readout_stages(128)[0]).split("-")[1], 16))
        digest = torch.stack([probe.sum(), probe.abs().max(), probe[:, ::97].sum(), el[0], torch.tensor(order_word, dtype=torch.float64)])
        gathered = [torch.empty_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        g = torch.stack(gathered)
        # the last step's gathered rows: rank r's block must be what rank r scored for ITS batch of that step
        step = args.warmup + args.steps - 1
        blocks_ok = all(torch.equal(last[r * bs:(r + 1) * bs],
                                    score(tasks.all_negative(data, triples[((step * world + r) * bs) % (triples.shape[0] - bs):][:bs])[0]))
                        for r in range(world))
        per_rank = {"ms_per_step": [1e3 * v / args.steps for v in g[:, 3].tolist()],
                    "probe_scores_identical": bool((g[:, :3] == g[0, :3]).all()),
                    "readout_order_id": ["order-%08x" % int(v) for v in g[:, 4].tolist()],
                    "readout_order_identical": bool((g[:, 4] == g[0, 4]).all()),
                    "gathered_rows_per_step": seen[-1][0], "gathered_blocks_in_rank_order": bool(blocks_ok)}
        # every rank's choice of slot streams (candidate, priority class, trial and settled figures)
        mine = {k: slot_report.get(k) for k in ("chosen", "candidate", "forced", "settle")} if slot_report else None
        if mine and slot_report.get("trial_ms"):
            ms = [m for _, m in slot_report["trial_ms"]]
            mine.update(chosen_ms=min(ms), spread_ms=[min(ms), max(ms)])
        choices = [None] * dist.get_world_size()
        dist.all_gather_object(choices, mine)
        per_rank["slot_streams"] = choices
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    out = {"metric": "triples scored/sec (all-tail ranking) on FB15k237", "value": world * bs * N * args.steps / elapsed,
           "unit": "triples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "stub (CPU dry run of the %d-rank program over gloo: NOT a measurement)" % world,
           "config": {"workload": "stub scorer on a 512-node synthetic graph: the N-rank control flow of bench.py without GPUs",
                      "batch_per_gpu": bs, "triples_per_step_per_gpu": bs * N, "rccl_world_size": world if gather else 1,
                      "collective_backend": "gloo (CPU dry run)" if gather else None, "per_rank": per_rank,
                      "parallelism": "query-shard x%d + all-gather of scores" % world if world > 1 else "single process"}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if gather:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", default="fb15k237")
    ap.add_argument("--bs", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-graph", action="store_true", help="launch the forward eagerly instead of replaying its hipGraph")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="captured forwards replayed round-robin on as many streams (consecutive batches are independent); 1 = one "
                         "stream.  Three: measured with the slots' streams picked by trial (graph.pick_slot_streams), plain process "
                         "0.568 / 0.558 - 0.573 ms first run / repeats against 0.583 - 0.593 / 0.584 - 0.589 with two; a launcher's rank "
                         "0.565 - 0.588 / 0.555 - 0.572 against 0.600 - 0.650 / 0.585 - 0.643 (profiles/r5_experiments.txt)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes of the roofline block")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fine-tuning steps of the `secondary` block")
    ap.add_argument("--data-root", default=None,
                    help="directory with train.txt / valid.txt / test.txt (kg-datasets/FB15k-237 layout): score the real test "
                         "triples instead of the synthetic graph of --shape (data: \"real\")")
    ap.add_argument("--pre-warm", type=int, default=64,
                    help="untimed steps in front of the first timed run's W warm-up steps (the chip's clocks settle after tens of ms)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed loop of --steps steps runs this many times in the process: ms_per_step / value are the FIRST "
                         "run's (the contract's exactly-K-steps figure), `repeats` carries every run, their median and spread")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl = RCCL (the measured configuration).  gloo: the N-rank code path where ranks must share a GPU "
                         "(tests on a one-GPU box: RCCL refuses two ranks on one device); ranks then map to GPUs modulo the "
                         "visible count and the line says so")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="dry run of the N-rank program WITHOUT GPUs (tests/test_bench_cpu.py): the launcher path, the gloo "
                         "process group, the shared readout order, the per-step all-gather behind each pipeline slot, the "
                         "barrier-fenced max-over-ranks clock and the JSON line, around a stub scorer on CPU -- not a measurement")
    ap.add_argument("--pmc-target", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--trace-target", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.pmc_target or args.trace_target:
        import __graft_entry__ as entry
        entry.build()
        pmc_target() if args.pmc_target else trace_target()
        return

    import torch.distributed as dist
    if args.stub_cpu:
        return stub_main(args)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as a plain `python bench.py --gpus N`: launch the N ranks ourselves, the way the driver does (one process per
        # GPU over RCCL), or refuse -- never report one GPU's number as N GPUs'
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) are visible; refusing to benchmark fewer GPUs than asked for"
                     % (args.gpus, have))
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        if args.backend != "nccl":
            sys.exit("bench.py: --backend gloo is for launcher-started test runs")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d does not match the launcher's world size %d" % (args.gpus, world))
    if local_rank >= torch.cuda.device_count() and args.backend == "nccl":
        sys.exit("bench.py: rank %d has no GPU (local rank %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
    gpu_index = local_rank % torch.cuda.device_count()       # (gloo test mode: ranks may share a GPU)
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # started by torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout carries exactly one JSON
        # line here, so the banner goes to stderr (file-descriptor level: it is written by C code)
        with _stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)           # "nccl" is RCCL on ROCm
            else:
                dist.init_process_group("gloo")
            dist.barrier()                                                # (communicator creation happens here)
            torch.cuda.synchronize()

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    if UPDATE_FORM:
        from ultra_amd import rspmm as _r
        _r.set_tuning(update_form=UPDATE_FORM)
    from ultra_amd import distributed as udist
    from ultra_amd import host_order, models, rspmm, synthetic, tasks
    if world > 1 or launched:
        # one association of the readout's last product for the whole job: rank 0 resolves it (cache / probe), the others adopt
        # its program, the ids are compared (no per-rank probe, no cache race)
        udist.share_readout_order(128, device=dev)

    data_kind, data_name = "synthetic", "%s-shaped synthetic KG" % args.shape
    if args.data_root:
        # real triples when the files are provided
        from ultra_amd.data import load_triples_dir
        data_cpu = load_triples_dir(args.data_root)
        data_kind, data_name = "real", "%s (raw triples, test split)" % os.path.basename(os.path.normpath(args.data_root))
    else:
        data_cpu = synthetic.make_kg(**synthetic.SHAPES[args.shape], seed=1234)
    data = data_cpu.to(dev)
    cfg = synthetic.default_model_cfg()
    torch.manual_seed(0)
    model = models.Ultra(**cfg)
    weights = "random-init"
    golden = os.path.join(ROOT, "tests", "golden", "ultra_3g_model.pt")
    if os.path.exists(golden):
        model.load_state_dict(torch.load(golden))
        weights = "ultra_3g state dict (tests/golden fixture)"
    model = model.to(dev).eval()

    N = data.num_nodes
    bs = args.bs
    triples = data.target_triples     # (num_test, 3) on device

    def batch_for(step):
        lo = ((step * world + rank) * bs) % (triples.shape[0] - bs)
        return triples[lo:lo + bs]

    def eager_forward(data_, batch_, post=None):
        score = model(data_, batch_)
        return post(score) if post is not None else score

    slot_report = {}      # which streams the pipeline slots of the timed forward run on (graph.pick_slot_streams)
    pipelines = []        # the PipelinedForward behind `forward` (its settle() watches the pre-warm steps)

    def make_forward():
        if args.no_graph:
            return eager_forward
        # the ~30-launch forward is captured once into a hipGraph and replayed (ultra_amd/graph.py); every step
        # still scores a fresh batch: its candidates are copied into the graph's input buffer first.  With --in-flight 2
        # (default) two such captures take the batches alternately on two streams: the launches of one batch that leave the
        # chip idle -- relation model, glue -- run beside the entity layers of its neighbour (graph.PipelinedForward).
        from ultra_amd.graph import GraphedForward, PipelinedForward
        try:
            example = tasks.all_negative(data, batch_for(0))[0]
            if args.in_flight > 1 and rspmm._plan_defaults["exact_order"]:
                piped = PipelinedForward(model, data, example, depth=args.in_flight,
                                         trial_post=udist.all_gather_scores if (world > 1 or launched) else None)
                if not slot_report:
                    slot_report.update(piped.stream_report or {})
                pipelines.append(piped)
                return lambda data_, batch_, post=None: piped(batch_, post=post)
            graphed = GraphedForward(model, data, example)

            def graphed_forward(data_, batch_, post=None):
                score = graphed(batch_)
                return post(score) if post is not None else score
            return graphed_forward
        except Exception as exc:      # capture refused by the runtime: the same forward, launched eagerly
            print("[bench] hipGraph capture failed (%s); running eagerly" % exc, file=sys.stderr)
            torch.cuda.synchronize()
            args.no_graph = True
            return eager_forward

    # synthetic input, resident in HBM before the timed region: one (bs, N, 3) all-tail candidate batch per step
    # (distinct queries per step; cycled beyond 256 steps)
    n_inputs = min(args.warmup + args.steps, 256)
    inputs = [tasks.all_negative(data, batch_for(i))[0] for i in range(n_inputs)]

    pre_warm = [max(args.pre_warm, 0)]      # (consumed by the first timed run)

    def timed_run(forward, gather):
        def one_step(step):
            # (bs, N) scores; with `gather` one RCCL all-gather per step, enqueued right behind the forward on its stream:
            # (world * bs, N)
            return forward(data, inputs[step % n_inputs], post=udist.all_gather_scores if gather else None)
        with torch.no_grad():
            # the chip reaches its sustained clocks only after some tens of milliseconds of work (measured: the first 20 steps after
            # 5 warm-up steps run 3 - 4 % slower than the same 20 steps repeated; after 50 warm-up steps they do not) -- so the
            # FIRST timed run of the process is preceded by `--pre-warm` untimed steps (reported as `pre_warmup_steps`), then
            # the W warm-up steps and the K timed steps as asked for
            n_pre = pre_warm.pop() if pre_warm else 0
            if n_pre and pipelines:
                # the pre-warm steps double as the check of the slot streams' choice against the steady state: slower than the
                # trial promised by more than 3 % -> one more choice (graph.PipelinedForward.settle)
                settled = pipelines[0].settle(one_step, steps=n_pre, collective=gather)
                if settled is not None:
                    slot_report["settle"] = settled
            else:
                for i in range(n_pre):
                    one_step(i)
            for i in range(args.warmup):
                one_step(i)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                one_step(args.warmup + i)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

    roofline = None
    if rank == 0 and world == 1 and not launched and not args.no_roofline:
        # the dominant kernel's own measurement (HIP events around its launches, rocprofv3 child passes) runs BEFORE the timed
        # region: it is independent of it, and the GPU then enters the timed steps from sustained work rather than from idle
        roofline = measure_roofline(dev, use_pmc=not args.no_pmc, in_flight=1 if args.no_graph else args.in_flight)
    forward = make_forward()
    elapsed = timed_run(forward, world > 1 or launched)
    # the same K steps again, `--repeats` runs in all: box-to-box and run-to-run spread is of the size of a small kernel gain,
    # so the line carries the median and the extremes beside the first run's figure
    runs = [elapsed] + [timed_run(forward, world > 1 or launched) for _ in range(max(args.repeats, 1) - 1)]
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    per_rank = None
    if world > 1 or launched:
        # per-rank consistency: every rank scores the SAME probe batch on its own GPU (its own plan upload, its own capture);
        # the digests and the ranks' own clocks are gathered over RCCL and compared on rank 0
        with torch.no_grad():
            probe = model(data, tasks.all_negative(data, triples[:bs])[0]).double()
        order_word = float(int(host_order.order_id(host_order.readout_stages(128)[0]).split("-")[1], 16))
        digest = torch.stack([probe.sum(), probe.abs().max(), probe[:, ::97].sum(), el[0],
                              torch.tensor(order_word, device=dev, dtype=torch.float64)])
        gathered = [torch.empty_like(digest) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, digest)
        g = torch.stack(gathered).cpu()
        per_rank = {"ms_per_step": [1e3 * v / args.steps for v in g[:, 3].tolist()],
                    "probe_scores_identical": bool((g[:, :3] == g[0, :3]).all()),
                    "readout_order_id": ["order-%08x" % int(v) for v in g[:, 4].tolist()],
                    "readout_order_identical": bool((g[:, 4] == g[0, 4]).all())}
        # every rank's choice of slot streams (candidate, priority class, trial and settled figures)
        mine = {k: slot_report.get(k) for k in ("chosen", "candidate", "forced", "settle")} if slot_report else None
        if mine and slot_report.get("trial_ms"):
            ms = [m for _, m in slot_report["trial_ms"]]
            mine.update(chosen_ms=min(ms), spread_ms=[min(ms), max(ms)])
        choices = [None] * dist.get_world_size()
        dist.all_gather_object(choices, mine)
        per_rank["slot_streams"] = choices
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    triples_per_s = world * bs * N * args.steps / elapsed

    out = {
        "metric": "triples scored/sec (all-tail ranking) on FB15k237",
        "value": triples_per_s, "unit": "triples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "pre_warmup_steps": max(args.pre_warm, 0),
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "ms_per_step_median": 1e3 * sorted(runs)[len(runs) // 2] / args.steps,
        "repeats": {"runs": len(runs), "steps_per_run": args.steps,
                    "ms_per_step": [round(1e3 * r / args.steps, 5) for r in runs],
                    "median": 1e3 * sorted(runs)[len(runs) // 2] / args.steps,
                    "min": 1e3 * min(runs) / args.steps, "max": 1e3 * max(runs) / args.steps,
                    "note": "this rank's clock; `ms_per_step` / `value` above are the first run (max over ranks)"},
        "vs_baseline": None, "dtype": "f32", "data": data_kind,
        "config": {"workload": "ultra_3g architecture zero-shot all-tail ranking, %s "
                               "(N=%d, E=%d, R=%d), distmult+sum rspmm, batch %d queries/GPU, query-sharded"
                               % (data_name, N, data.num_edges, data.num_relations, bs),
                   "batch_per_gpu": bs, "triples_per_step_per_gpu": bs * N, "weights": weights,
                   # (what ran in front of the K timed steps of `ms_per_step`: W warm-up steps as asked for, behind
                   # `pre_warmup_steps` untimed steps of the same loop; the sustained figure is the median of the repeats)
                   "untimed_steps_before_the_first_run": {"warmup": args.warmup, "pre_warmup_steps": max(args.pre_warm, 0)},
                   "ms_per_step_median_of_repeats": 1e3 * sorted(runs)[len(runs) // 2] / args.steps,
                   "summation_order": "reference (rspmm.cpp:61-72 sequential per row; nn.Linear / nn.LayerNorm in torch's CPU order; "
                                      "readout GEMV %s)" % host_order.describe(128),
                   "readout_order_id": host_order.order_id(host_order.readout_stages(128)[0]),
                   "launch": ("eager" if args.no_graph else
                              "hipGraph replay of the captured forward" +
                              (", %d batches in flight on %d streams, the aggregation kernels of each on three quarters of the CUs "
                               "(graph.PipelinedForward)" % (args.in_flight, args.in_flight)
                               if args.in_flight > 1 else "")),
                   "hip_hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"),
                   # (normal- or high-priority pair, whichever interleaved better in a ~ 20 ms trial when the pipeline was built)
                   "slot_streams": slot_report or None,
                   "rccl_world_size": dist.get_world_size() if (world > 1 or launched) else 1,
                   "collective_backend": ("RCCL" if args.backend == "nccl" else "gloo (test mode: ranks share GPUs, not a measurement)")
                   if (world > 1 or launched) else None,
                   "per_rank": per_rank,
                   "parallelism": "query-shard x%d + RCCL all-gather of scores" % world if world > 1 else "single GPU"},
    }

    if rank == 0 and not args.no_roofline:
        # (counter passes only at N = 1: they re-run the kernels in a child process on this rank's GPU)
        out["roofline"] = roofline if roofline is not None else measure_roofline(dev, use_pmc=False,
                                                                               in_flight=1 if args.no_graph else args.in_flight)
    if rank == 0 and world == 1:

        # ---- CPU baseline + parity on the identical batch ----
        if not args.no_cpu_baseline:
            from oracle import ultra_oracle_model
            fn = ultra_oracle_model.reference_rspmm_fn()
            ncores = available_cores()
            host_threads = torch.get_num_threads()
            torch.set_num_threads(ncores)
            batch = data_cpu.target_triples[:bs]
            t_batch_cpu, _ = tasks.all_negative(data_cpu, batch)
            state = {k: v.cpu() for k, v in model.state_dict().items()}
            ultra_oracle_model.ultra_forward(state, cfg, data_cpu, t_batch_cpu, rspmm_fn=fn)      # warm-up
            n_fwd, t_cpu0 = 0, time.perf_counter()
            while True:
                ref_score = ultra_oracle_model.ultra_forward(state, cfg, data_cpu, t_batch_cpu, rspmm_fn=fn)
                n_fwd += 1
                if time.perf_counter() - t_cpu0 >= args.cpu_seconds or n_fwd >= 20:
                    break
            t_cpu = time.perf_counter() - t_cpu0
            cpu_model = "unknown"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
            except Exception:
                pass
            out["cpu_baseline"] = {"value": bs * N * n_fwd / t_cpu, "unit": "triples/s", "cores": ncores,
                                   "kind": "port",
                                   "sample": "%d all-tail forwards of the same %d-query batch through oracle/ultra_oracle_model.py "
                                             "(torch CPU ops in the reference's data flow; rspmm = %s), %.1f s"
                                             % (n_fwd, bs, "the reference's own rspmm.cpp TU via oracle/_ref"
                                                if fn is not None else "C oracle rspmm", t_cpu),
                                   "cpu": cpu_model, "ms_per_forward": 1e3 * t_cpu / n_fwd,
                                   "note": "`port`: /root/reference does not exist on the GPU box, so the unchanged ultra.models.Ultra "
                                           "cannot be imported here; the restatement is pinned to it by tests/golden (recorded from the "
                                           "unchanged reference modules).  Survey container (8 vCPU Xeon 2.1 GHz), unchanged reference "
                                           "Ultra.forward: 910 ms per forward = 0.128 M triples/s (SURVEY.md section 6)."}
            # op level: the reference kernel on the identical pre-sorted tensors, min of 3 after a warm-up
            try:
                from oracle import build_ref, rspmm_oracle
                if build_ref.available():
                    ref_mod = build_ref.load()
                    gen = torch.Generator().manual_seed(0)
                    ei_s, et_s, ew_s, _ = rspmm_oracle.sort_edges(data_cpu.edge_index, data_cpu.edge_type,
                                                                   torch.ones(data_cpu.num_edges))
                    rel_cpu = torch.randn(data_cpu.num_relations, bs * 64, generator=gen)
                    x_cpu = torch.randn(N, bs * 64, generator=gen)
                    op_ms = {}
                    for name in ("rspmm_add_mul_forward_cpu", "rspmm_max_mul_forward_cpu"):
                        f = getattr(ref_mod, name)
                        f(ei_s, et_s, ew_s, rel_cpu, x_cpu)
                        best = float("inf")
                        for _ in range(3):
                            t0 = time.perf_counter()
                            f(ei_s, et_s, ew_s, rel_cpu, x_cpu)
                            best = min(best, time.perf_counter() - t0)
                        op_ms[name] = 1e3 * best
                    out["cpu_baseline"]["reference_kernel_ms"] = op_ms
            except Exception as exc:       # the op-level figure is context only
                out["cpu_baseline"]["reference_kernel_ms"] = "unavailable: %s" % exc
            with torch.no_grad():
                got = model(data, t_batch_cpu.to(dev)).cpu()
            # fp64 run of the same oracle: the value both fp32 implementations approximate
            truth = ultra_oracle_model.ultra_forward({k: v.double() for k, v in state.items()}, cfg, data_cpu, t_batch_cpu)
            t_mask, _ = tasks.strict_negative_mask(data_cpu, batch)
            pos_t = batch[:, 1]
            r_gpu = tasks.compute_ranking(got, pos_t, t_mask)
            r_cpu = tasks.compute_ranking(ref_score, pos_t, t_mask)
            r_true = tasks.compute_ranking(truth.float(), pos_t, t_mask)
            diff = (got - ref_score).abs().max().item()
            out["parity"] = {"max_abs_score_diff": diff, "tolerance": 1e-4, "margin": 1e-4 / max(diff, 1e-30),
                             "rank_mismatches": int((r_gpu != r_cpu).sum()), "queries": bs,
                             "rank_mismatches_outside_reference_ties": tie_band_mismatches(ref_score, r_gpu, pos_t, t_mask, 2 * diff),
                             "tie_band": "a rank differs `outside reference ties` when no shift of the positive's reference score "
                                         "by <= 2 * max_abs_score_diff reproduces it",
                             "max_abs_err_gpu_vs_fp64": (got.double() - truth).abs().max().item(),
                             "max_abs_err_reference_fp32_vs_fp64": (ref_score.double() - truth).abs().max().item(),
                             "rank_mismatches_gpu_vs_fp64": int((r_gpu != r_true).sum()),
                             "rank_mismatches_reference_fp32_vs_fp64": int((r_cpu != r_true).sum()),
                             "scores_bit_equal": int((got == ref_score).sum()), "scores": got.numel(),
                             "metrics_gpu": ranking_metrics(r_gpu), "metrics_reference": ranking_metrics(r_cpu),
                             "readout_order": host_order.describe(128),
                             "note": "every operation of the forward follows the reference's order (rspmm.cpp row sums, torch's "
                                     "nn.Linear / nn.LayerNorm arithmetic, the host BLAS's association for the readout's last "
                                     "product, probed by ultra_amd/host_order.py); the host BLAS may sum a few trailing rows of "
                                     "each thread's share with a remainder kernel"}
            # ---- Hits@k where it is not vacuous: 16 fact-graph edges whose reference ranks are 1 .. 30 (tests/golden) ----
            topk_file = os.path.join(ROOT, "tests", "golden", "topk_queries_fb15k237.json")
            if args.shape == "fb15k237" and not args.data_root and os.path.exists(topk_file):
                rec = json.load(open(topk_file))
                fact = torch.stack([data_cpu.edge_index[0], data_cpu.edge_index[1], data_cpu.edge_type], dim=-1)
                queries = fact[torch.tensor(rec["indices"])]
                rg, rr = [], []
                for b0 in range(0, len(queries), bs):
                    qb = queries[b0:b0 + bs]
                    cand, _ = tasks.all_negative(data_cpu, qb)
                    qmask, _ = tasks.strict_negative_mask(data_cpu, qb)
                    want_q = ultra_oracle_model.ultra_forward(state, cfg, data_cpu, cand, rspmm_fn=fn)
                    with torch.no_grad():
                        got_q = model(data, cand.to(dev)).cpu()
                    rg.append(tasks.compute_ranking(got_q, qb[:, 1], qmask))
                    rr.append(tasks.compute_ranking(want_q, qb[:, 1], qmask))
                rg, rr = torch.cat(rg), torch.cat(rr)
                out["parity"]["top_ranked_queries"] = {
                    "queries": len(queries), "source": "fact-graph edges with reference ranks 1 .. 30 (tests/golden/topk_queries_fb15k237.json)",
                    "metrics_gpu": ranking_metrics(rg), "metrics_reference": ranking_metrics(rr),
                    "rank_mismatches": int((rg != rr).sum()), "metrics_identical": ranking_metrics(rg) == ranking_metrics(rr)}
            torch.set_num_threads(host_threads)      # (the CPU baseline's thread count must not leak into the host-side work below)
            # ---- the re-associating plans (round 1's timed path), same command: throughput and parity beside the timed mode ----
            rspmm.set_plan_defaults(exact_order=False)
            try:
                el2 = timed_run(make_forward(), False)
                with torch.no_grad():
                    got2 = model(data, t_batch_cpu.to(dev)).cpu()
                r2 = tasks.compute_ranking(got2, pos_t, t_mask)
                out["modes"] = {
                    "reference_order": {"timed": True, "triples_per_s": triples_per_s, "ms_per_step": out["ms_per_step"],
                                        "rank_mismatches": out["parity"]["rank_mismatches"], "max_abs_score_diff": diff},
                    "reassociated": {"timed": False, "triples_per_s": bs * N * args.steps / el2,
                                     "ms_per_step": 1e3 * el2 / args.steps,
                                     "rank_mismatches": int((r2 != r_cpu).sum()),
                                     "max_abs_score_diff": (got2 - ref_score).abs().max().item(),
                                     "note": "set_plan_defaults(exact_order=False): split hub rows + dense-format twins; sums "
                                             "re-associated, nn.Linear / nn.LayerNorm still in torch's order"}}
            finally:
                rspmm.set_plan_defaults()
            # ---- entity layers as TWO launches each (rspmm, then conv_update): the timed mode runs them as one ----
            from ultra_amd import layers as _layers
            was = _layers.FUSED_SPARSE_LAYER
            _layers.FUSED_SPARSE_LAYER = False
            try:
                el3 = timed_run(make_forward(), False)
                with torch.no_grad():
                    got3 = model(data, t_batch_cpu.to(dev)).cpu()
                out.setdefault("modes", {})["two_launch_layers"] = {
                    "timed": False, "triples_per_s": bs * N * args.steps / el3, "ms_per_step": 1e3 * el3 / args.steps,
                    "scores_bit_equal_with_the_timed_mode": bool(torch.equal(got3, got)),
                    "note": "ULTRA_FUSED_SPARSE_LAYER=0: ultra_rspmm_forward_point + ultra_conv_update per entity layer instead of "
                            "ultra_rspmm_forward_update; same bits (DESIGN.md 3.8)"}
            finally:
                _layers.FUSED_SPARSE_LAYER = was
            # ---- one batch at a time: the same captured forward on one stream ----
            if args.in_flight > 1 and not args.no_graph:
                keep = args.in_flight
                args.in_flight = 1
                try:
                    el4 = timed_run(make_forward(), False)
                    out.setdefault("modes", {})["one_batch_in_flight"] = {
                        "timed": False, "triples_per_s": bs * N * args.steps / el4, "ms_per_step": 1e3 * el4 / args.steps,
                        "note": "--in-flight 1: every batch waits for its predecessor's readout (DESIGN.md 3.9)"}
                finally:
                    args.in_flight = keep
                # ---- the batches in flight with whole-chip launches (round 3's form of the timed mode) ----
                try:
                    from ultra_amd.graph import PipelinedForward
                    piped = PipelinedForward(model, data, tasks.all_negative(data, batch_for(0))[0], depth=args.in_flight, share_chip=False)
                    el6 = timed_run(lambda data_, batch_, post=None: piped(batch_, post=post), False)
                    out.setdefault("modes", {})["in_flight_whole_chip_launches"] = {
                        "timed": False, "triples_per_s": bs * N * args.steps / el6, "ms_per_step": 1e3 * el6 / args.steps,
                        "note": "PipelinedForward(share_chip=False): every aggregation kernel with one workgroup per CU, so the entity "
                                "layers of the batches in flight follow one another (DESIGN.md 3.9)"}
                    del piped
                except Exception as exc:
                    out.setdefault("modes", {})["in_flight_whole_chip_launches"] = {"unavailable": str(exc)[:200]}
            # ---- NOT the timed mode: the relation model's output for every query relation computed once (it depends on the
            # query relation only; Ultra.cache_relation_representations, what evaluate() does for long shards) ----
            try:
                model.cache_relation_representations(data, chunk=bs)
                el5 = min(timed_run(make_forward(), False) for _ in range(2))
                with torch.no_grad():
                    got5 = model(data, t_batch_cpu.to(dev)).cpu()
                out.setdefault("modes", {})["relation_table"] = {
                    "timed": False, "triples_per_s": bs * N * args.steps / el5, "ms_per_step": 1e3 * el5 / args.steps,
                    "scores_bit_equal_with_the_timed_mode": bool(torch.equal(got5, got)),
                    "note": "work the timed step does is SKIPPED here (the relation model runs once per relation, not once per "
                            "step): an engine feature for jobs that score many batches over one graph, not a benchmark figure"}
            finally:
                model.drop_relation_cache()
    if rank == 0 and world == 1 and not launched and not args.no_secondary:
        # ---- BASELINE.json config 5 (fine-tuning): fwd + bwd + AdamW per step, beside the headline ----
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import secondary_bench
        del model, data
        rspmm.clear_plan_cache()
        torch.cuda.empty_cache()
        out["secondary"] = {
            # BASELINE.json configs 3 and 1 (config/transductive/inference.yaml:13-14,21-22; ultra/layers.py:206-207)
            "forward": [secondary_bench.forward_parity_case("codex_l", "max", "ultra_50g", bs=8),
                        secondary_bench.forward_parity_case("codex_l", "sum", "ultra_50g", bs=8),
                        secondary_bench.forward_parity_case("wn18rr", "sum", "ultra_3g", bs=4)],
            "fine_tune": [secondary_bench.train_case("fb15k237"), secondary_bench.train_case("yago310"),
                          secondary_bench.train_case("fb15k237", aggr="max")],
            "sparse_relation_graph": secondary_bench.sparse_relation_case("fb15k237", fill=0.12),
            # the full test() protocol (script/run.py:121-226) as a secondary number -- all 20,466 test triples
            "evaluate": secondary_bench.evaluate_case("fb15k237", "ultra_3g", bs=8, in_flight=3),
            "note": "one optimisation step of script/run.py:40-90 (strict negatives, train()-mode forward with the batch's own "
                    "edges dropped, self-adversarial BCE, backward, AdamW) on synthetic graphs of the named shapes; batch 8 x "
                    "(1 + 256 negatives).  rspmm forward / backward and the layer update (forward and backward) run on the HIP "
                    "engine; `ms_per_step` is the step as ONE hipGraph replay (ultra_amd/train.py: 20 timed steps after 3), "
                    "`ms_per_step_eager` the same step launched kernel by kernel (10 timed steps after 3)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
