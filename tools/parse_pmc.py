"""Turn the two rocprofv3 --pmc passes of tools/roofline_points.py (FETCH_SIZE, WRITE_SIZE; separate runs) into
HBM-side bytes per launch.  Calibration (MI355X_MICROARCH.md, HBM section): the streaming copy of a known
1 GiB fixes the unit of each counter in THIS environment; the factors are stored next to the results."""
import csv
import json
import sys

POINTS = [("fb15k237", 8), ("fb15k237", 64), ("codex_l", 8), ("yago310", 8)]
COPY_BYTES = 1 << 30


def per_kernel(path, counter):
    seq = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = "copy" if "stream_copy_kernel" in name else ("fwd_add" if "rspmm_fwd_kernel<float, 4, 0, 0" in name else (
            "fwd_max" if "rspmm_fwd_kernel<float, 4, 2, 0" in name else (
                "fix_add" if "rspmm_fixup_kernel<float, 4, 0>" in name else (
                    "fix_max" if "rspmm_fixup_kernel<float, 4, 2>" in name else None))))
        if key:
            seq.setdefault(key, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in seq.items()}


def main(fetch_csv, write_csv, out):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    f_unit = COPY_BYTES / (sum(f["copy"]) / len(f["copy"]))      # bytes per counter unit, calibrated
    w_unit = COPY_BYTES / (sum(w["copy"]) / len(w["copy"]))
    res = {"calibration": {"copy_bytes": COPY_BYTES, "FETCH_SIZE_mean": sum(f["copy"]) / len(f["copy"]),
                           "WRITE_SIZE_mean": sum(w["copy"]) / len(w["copy"]), "fetch_bytes_per_unit": f_unit,
                           "write_bytes_per_unit": w_unit,
                           "note": "FETCH_SIZE counts 1 KiB units at half rate for 16-B/lane streams on gfx950 (x2 correction); "
                                   "WRITE_SIZE is exact in KiB"},
           "points": []}
    # launches per (point, sum) in --pmc mode: 1 warm-up + 3 back-to-back + 3 kernel-only timed calls
    per = len(f["fwd_add"]) // len(POINTS)
    assert per * len(POINTS) == len(f["fwd_add"]) == len(w["fwd_add"]), "unexpected dispatch count"
    for i, (shape, bs) in enumerate(POINTS):
        for sum_ in ("add", "max"):
            fk, xk = "fwd_" + sum_, "fix_" + sum_
            lo, hi = i * per, (i + 1) * per
            fetch = sum(f[fk][lo:hi]) / per * f_unit + sum(f[xk][lo:hi]) / per * f_unit
            write = sum(w[fk][lo:hi]) / per * w_unit + sum(w[xk][lo:hi]) / per * w_unit
            res["points"].append({"shape": shape, "bs": bs, "sum": sum_, "hbm_read_bytes_per_launch": fetch,
                                  "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write})
    bench_pt = res["points"][0]
    res["hbm_bytes_per_launch"] = bench_pt["hbm_bytes_per_launch"]      # bench.py's roofline.traffic (fb15k237, bs 8, add)
    json.dump(res, open(out, "w"), indent=1)
    for p in res["points"]:
        print("%-9s bs=%-3d %s  read %.1f MB  write %.1f MB  total %.1f MB" %
              (p["shape"], p["bs"], p["sum"], p["hbm_read_bytes_per_launch"] / 1e6, p["hbm_write_bytes_per_launch"] / 1e6,
               p["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
