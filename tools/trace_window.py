"""Prints every kernel of both streams between two main-stream kernels of the second-to-last step of a rocprofv3 kernel trace
(who runs while the main stream idles).  Usage: python tools/trace_window.py <kernel_trace.csv> <from-kernel> <to-kernel>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seg = rows[adam[-3] + 1:adam[-2] + 1]
main_id = seg[-1]["Stream_Id"]
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
i0 = next(i for i, r in enumerate(seg) if sys.argv[2] in r["Kernel_Name"])
i1 = next(i for i, r in enumerate(seg) if i > i0 and sys.argv[3] in r["Kernel_Name"])
t0 = int(seg[i0]["Start_Timestamp"])
for r in seg[max(0, i0 - 3):i1 + 3]:
    print("%9.1f us  +%7.1f us  %-5s %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          "main" if r["Stream_Id"] == main_id else "s" + r["Stream_Id"], nm(r), r.get("Grid_Size", "")))
