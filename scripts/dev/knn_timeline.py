"""Dev tool: print the in-situ timeline of the KNN kernels from a rocprofv3 --kernel-trace CSV.

usage: python scripts/dev/knn_timeline.py <kernel_trace.csv> [out.txt]
"""
import collections
import csv
import sys


def short(n):
    for k in ("prep", "split", "filter", "refine", "scatter"):
        if k in n:
            return k
    return n[:10]


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1]))
            if "knn_" in r["Kernel_Name"] or "ratio_scatter" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sel = rows[-900:-400]          # steady state: before the isolated profiling samples at the end
    t0 = int(sel[0]["Start_Timestamp"])
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    for r in sel[:60]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        out.write("%-8s q%3s start %9.1f us  dur %7.1f\n" % (short(r["Kernel_Name"]), r["Queue_Id"], s / 1e3, (e - s) / 1e3))
    dur = collections.defaultdict(list)
    for r in sel:
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        out.write("%s: n %d mean %.1f us\n" % (k, len(v), sum(v) / len(v)))
    span = (int(sel[-1]["End_Timestamp"]) - t0) / 1e3
    out.write("window %.1f us for %d kernels\n" % (span, len(sel)))


main()
