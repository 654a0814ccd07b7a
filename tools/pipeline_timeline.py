"""What the chip does during the pipelined benchmark step, from a `rocprofv3 --kernel-trace` CSV of tools/step_probe.py (or bench.py):
the last `window_ms` of the trace, per stream the kernels in time order (start offset, duration, workgroups), and for the entity-layer
kernel (rspmm_order_kernel): how much of the window has 0 / 1 / 2+ of its launches running, its mean duration, and the gaps between the
end of one entity launch and the start of the next one on ANY stream.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python tools/step_probe.py 3 60
    python tools/pipeline_timeline.py /tmp/pt/*/*_kernel_trace.csv [window_ms] [--list]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"at::native::|\(anonymous namespace\)::|void |ultra::", "", name)
    return re.sub(r"\(.*", "", name)[:60]


def main(path, window_ms=3.0, listing=False):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    end = max(r["e"] for r in rows)
    lo = end - int(window_ms * 1e6)
    win = sorted((r for r in rows if r["s"] >= lo), key=lambda r: r["s"])
    ent = [r for r in win if "rspmm_order_kernel" in r["Kernel_Name"]]
    if listing:
        for r in win:
            wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            print("%9.1f us  %7.1f us  stream %-3s queue %-3s wg %5d  %s" % ((r["s"] - lo) / 1e3, (r["e"] - r["s"]) / 1e3, r["Stream_Id"],
                                                                          r["Queue_Id"], wg, short(r["Kernel_Name"])))
    # coverage of the window by entity launches
    events = sorted([(r["s"], 1) for r in ent] + [(r["e"], -1) for r in ent])
    level, last, cover = 0, lo, {0: 0, 1: 0, 2: 0}
    for t, d in events:
        cover[min(level, 2)] += t - last
        level += d
        last = t
    cover[min(level, 2)] += end - last
    total = end - lo
    print("window %.2f ms: %d launches (%d entity layers), entity layers running: none %.1f %%, one %.1f %%, two or more %.1f %%"
          % (total / 1e6, len(win), len(ent), 100 * cover[0] / total, 100 * cover[1] / total, 100 * cover[2] / total))
    if ent:
        d = sorted((r["e"] - r["s"]) / 1e3 for r in ent)
        print("entity layer duration: mean %.1f us, median %.1f, min %.1f, max %.1f; %.1f per ms of window"
              % (sum(d) / len(d), d[len(d) // 2], d[0], d[-1], len(ent) / (total / 1e6)))
    by = {}
    for r in win:
        a = by.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += r["e"] - r["s"]
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        print("%5d x %8.1f us mean  %5.1f %% of the window  %s" % (n, t / n / 1e3, 100 * t / total, k))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(args[0], float(args[1]) if len(args) > 1 else 3.0, "--list" in sys.argv)
