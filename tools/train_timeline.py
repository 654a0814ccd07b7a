"""One fine-tuning step's launches in order, from a `rocprofv3 --kernel-trace` CSV (…_kernel_trace.csv): per launch the kernel
name (shortened), its duration and the idle gap in front of it; then busy time, gap time and the launch count of the step.
A step ends with the optimiser's multi-tensor kernels; the step printed is the last complete one.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python tools/train_graph_probe.py fb15k237
    python tools/train_timeline.py gpurun_out/tl/*/*_kernel_trace.csv [--all-stream] [--summary]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"at::native::|\(anonymous namespace\)::|void |ultra::|rocprim::ROCPRIM_\d+_NS::detail::", "", name)
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"(vectorized_elementwise_kernel|elementwise_kernel_manual_unroll|elementwise_kernel)<.*?(\w+Functor\w*|\w+_kernel_cuda|direct_copy|FillFunctor)", name)
    if m:
        return "elementwise:" + m.group(2)
    return name[:90]


def main(path, summary=False):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"]]
    # group consecutive adam launches into step ends
    step_ends = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] != i + 1]
    if len(step_ends) < 3:
        raise SystemExit("fewer than three steps in the trace")
    lo, hi = step_ends[-3] + 1, step_ends[-2] + 1
    step = rows[lo:hi]
    t_prev = int(rows[lo - 1]["End_Timestamp"])
    busy = gap = 0
    agg = {}
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = max(0, s - t_prev)
        if not summary:
            print("%8.1f us  gap %7.1f  %s" % ((e - s) / 1e3, g / 1e3, short(r["Kernel_Name"])))
        busy += e - s
        gap += g
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += e - s
        t_prev = max(t_prev, e)
    print("---- by kernel")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%4d x %9.1f us  %s" % (n, t / 1e3, k))
    span = int(step[-1]["End_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])
    print("launches %d  busy %.1f us  gaps %.1f us  span %.1f us" % (len(step), busy / 1e3, gap / 1e3, span / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], summary="--summary" in sys.argv)
