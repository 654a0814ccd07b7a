"""Summarise rocprofv3 --pmc passes over tools/pmc_target.py into a small text table (profiles/)."""
import collections
import csv
import sys

NAMES = (("true, 3>(ultra::OrderParams)", "entity layer, 1 launch"), ("true, 1>(ultra::OrderParams)", "entity layer, tail form"), ("rspmm_order_kernel", "entity rspmm"), ("4, 0, 0, 1>(ultra::FwdParams)", "training walk, fwd / input-grad"), ("4, 0, 0, 0>(ultra::FwdParams)", "training walk, relation-grad"), ("rspmm_fwd_kernel", "entity rspmm (r1)"), ("conv_update_kernel", "entity update"),
         ("dense_order_layer_kernel", "relation layer"), ("dense_layer_kernel", "relation layer (r1)"), ("readout_kernel", "readout"),
         ("rspmm_fixup_kernel", "fix-up"), ("conv_update_bwd_fused_kernel", "update backward"), ("conv_update_bwd_reduce_kernel", "update bwd reduce"))
out = []
for path in sys.argv[1:]:
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        label = next((lab for key, lab in NAMES if key in r["Kernel_Name"]), None)
        if label is None:
            continue
        agg.setdefault((label, r["Counter_Name"]), []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for (g, c), v in agg.items():
        v = [x for _, x in sorted(v)]
        v = v[1:] or v            # drop the warm-up launch
        out.append("%-34s %-26s %16.0f   (mean of %d launches)" % (g, c, sum(v) / len(v), len(v)))
print("\n".join(out))
