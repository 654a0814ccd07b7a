"""Summarise rocprofv3 --pmc passes over tools/pmc_target.py into a small text table (profiles/)."""
import collections
import csv
import sys

out = []
for path in sys.argv[1:]:
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "rspmm_fwd" not in k:
            continue
        key = ("entity(REL_LDS)" if ", 0, 0, 1>" in k else "relation(ALL_LDS)", r["Counter_Name"])
        agg.setdefault(key, []).append(float(r["Counter_Value"]))
    for (g, c), v in agg.items():
        v = v[1:] or v            # drop the warm-up launch
        out.append("%-18s %-24s %16.0f   (mean of %d launches)" % (g, c, sum(v) / len(v), len(v)))
print("\n".join(out))
