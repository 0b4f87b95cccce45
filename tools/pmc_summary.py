"""Average per-dispatch PMC counters of the conv MFMA kernels from a rocprofv3 --pmc counter_collection.csv."""
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_igemm_kernel" in k or "conv_wgrad_kernel" in k:
            short = k.split("::")[-1].split("(")[0] + " g" + r["Grid_Size"]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c, x in sorted(agg[k].items()):
        print("    %-28s %14.3f M  (n=%d)" % (c, sum(x) / len(x) / 1e6, len(x)))
