"""Dev: kernel timeline of the pipelined KNN step from a rocprofv3 --kernel-trace CSV (which kernels overlap, where the gaps are)."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = "filter" if "filter" in n else "refine" if "refine" in n else "prep" if "prep" in n else "scatter" if "scatter" in n else None
    if k:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
rows = rows[len(rows) // 3:]                      # steady state
t0 = rows[0][0]
fil = [r for r in rows if r[2] == "filter"]
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(fil, fil[1:])]
s2s = [(b[0] - a[0]) / 1e3 for a, b in zip(fil, fil[1:])]
dur = collections.defaultdict(list)
for s, e, k, q in rows:
    dur[k].append((e - s) / 1e3)
import statistics as st
print("kernel durations (us, median) under pipelining:", {k: round(st.median(v), 1) for k, v in dur.items()})
print("filter start-to-start us: median %.1f  | gap between consecutive filters (end -> next start): median %.1f min %.1f max %.1f" % (st.median(s2s), st.median(gaps), min(gaps), max(gaps)))
for s, e, k, q in rows[:24]:
    print("%8.1f %8.1f  %-8s q%s" % ((s - t0) / 1e3, (e - t0) / 1e3, k, q))
