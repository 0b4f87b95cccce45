import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
# last 150 ms of the run = steady-state steps
sel = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 200e6]
print("columns", list(rows[0].keys()))
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d > 150e3:
        print("%9.3f ms  dur %8.1f us  tid %s  %s" % ((int(r["Start_Timestamp"]) - t_end) / 1e6, d / 1e3, r.get("Thread_Id"), r["Function"]))
