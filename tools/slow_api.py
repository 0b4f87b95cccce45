"""Lists the HIP API calls of a rocprofv3 --hip-trace run (…_hip_api_trace.csv) that held the host for more than
<thr> us in the last <window> ms of the run: where the host blocks on the GPU."""
import csv, sys
from collections import Counter
rows = list(csv.DictReader(open(sys.argv[1])))
window = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) > t_end - window * 1e6]
tot = Counter(); cnt = Counter()
for i, r in enumerate(sel):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[r["Function"]] += d; cnt[r["Function"]] += 1
    if d > thr * 1e3:
        prev = sel[i - 1]["Function"] if i else "-"
        print("%9.3f ms  dur %8.1f us  tid %s  %s   (prev: %s)" % ((int(r["Start_Timestamp"]) - t_end) / 1e6, d / 1e3, r.get("Thread_Id"), r["Function"], prev))
print("totals in window:")
for k, v in tot.most_common(8):
    print("  %-28s %8.2f ms  %6d calls" % (k, v / 1e6, cnt[k]))
