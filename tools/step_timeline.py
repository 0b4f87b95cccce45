"""Reads a rocprofv3 kernel trace of bench.py (…_kernel_trace.csv) and prints, for the second-to-last training step,
the phases on the main stream (encoder forward / decode+loss / encoder backward / optimiser) with busy time, idle
gaps and the kernels that own them.  Usage: python tools/step_timeline.py <kernel_trace.csv> [top_n]"""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25


def nm(r):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0][:44]


adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seg = rows[adam[-3] + 1:adam[-2] + 1]
main = [r for r in seg if r["Stream_Id"] == seg[-1]["Stream_Id"]]
t0 = int(rows[adam[-3]]["End_Timestamp"])
print("step wall %.3f ms, %d kernels (%d on the main stream)" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e6, len(seg), len(main)))


def first(pred, start=0):
    return next(i for i in range(start, len(main)) if pred(nm(main[i])))


i_c1 = first(lambda n: "conv1_first_fwd" in n)
i_tp = first(lambda n: "time_pool" in n)
i_fb = first(lambda n: "freq_mean_bwd" in n)
i_cb = first(lambda n: "conv1_first_bwd" in n)
for label, lo, hi in (("host prep + bn0", 0, i_c1), ("encoder forward", i_c1, i_tp + 1), ("decode + loss + decode bwd", i_tp + 1, i_fb),
                      ("encoder backward", i_fb, i_cb + 1), ("clip + adam", i_cb + 1, len(main))):
    part = main[lo:hi]
    if not part:
        continue
    prev = t0 if lo == 0 else int(main[lo - 1]["End_Timestamp"])
    start = prev
    busy = gap = 0
    agg = defaultdict(lambda: [0, 0, 0])
    for r in part:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = max(0, st - prev)
        prev = max(prev, en)
        a = agg[nm(r) + " g" + r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"]]
        a[0] += 1; a[1] += en - st; a[2] += g
        busy += en - st; gap += g
    print("%-28s span %7.3f ms  busy %7.3f  idle %6.3f  n=%d" % (label, (prev - start) / 1e6, busy / 1e6, gap / 1e6, len(part)))
    for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:top]:
        print("     %-66s n=%4d busy=%8.1f us avg=%7.1f idle_before=%7.1f us" % (k, v[0], v[1] / 1e3, v[1] / v[0] / 1e3, v[2] / 1e3))

# idle gaps > 20 us on the main stream: what ran on the other stream meanwhile, and which kernel ended the gap
side = [r for r in seg if r["Stream_Id"] != seg[-1]["Stream_Id"]]
prev = None
print("gaps > 20 us on the main stream:")
for r in main:
    st = int(r["Start_Timestamp"])
    if prev is not None and st - int(prev["End_Timestamp"]) > 20e3:
        lo, hi = int(prev["End_Timestamp"]), st
        busy = [s for s in side if int(s["Start_Timestamp"]) < hi and int(s["End_Timestamp"]) > lo]
        print("  t=%8.3f ms gap %6.1f us  after %-28s before %-28s side-stream kernels in gap: %d (%s)" % (
            (lo - t0) / 1e6, (hi - lo) / 1e3, nm(prev)[:28], nm(r)[:28], len(busy), nm(busy[-1])[:24] if busy else "-"))
    if prev is None or int(r["End_Timestamp"]) > int(prev["End_Timestamp"]):
        prev = r
