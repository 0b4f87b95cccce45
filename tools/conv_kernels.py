"""Per-launch duration of every conv MFMA kernel of one encoder fwd+bwd (rocprofv3 kernel trace of bench_encoder.py),
in launch order with the GFLOP of the launch.  Usage: python tools/conv_kernels.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
conv = [r for r in rows if "conv_igemm" in r["Kernel_Name"] or "conv_wgrad" in r["Kernel_Name"]]
per_iter = 21            # 7 fwd + 7 dgrad + 7 wgrad
n_iter = len(conv) // per_iter
seqs = defaultdict(list)
for i, r in enumerate(conv[-per_iter * (n_iter - 2):]):   # skip warm-up iterations
    seqs[i % per_iter].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
for i in range(per_iter):
    d = sorted(x for x, _ in seqs[i]); r = seqs[i][0][1]
    name = "igemm" if "igemm" in r["Kernel_Name"] else "wgr192" if "wgrad192" in r["Kernel_Name"] else "wgrad"
    kn = r["Kernel_Name"]
    tmpl = kn.split("<")[1].split(">")[0] if "<" in kn else ("bf16" if "bf16" in kn else "")
    if "bf16" in kn:
        name += "16"
    print("%2d %-6s<%-8s> grid %8sx%3sx%4s  median %8.1f us" % (i, name, tmpl, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], d[len(d) // 2] / 1e3))
