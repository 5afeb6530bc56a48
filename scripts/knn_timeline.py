"""Kernel timeline of the KNN step in steady state (two launch sets in flight): from a rocprofv3 --kernel-trace csv, the last N
dispatches with start / end relative to the first, the queue they ran on, and per kernel the time it overlapped another queue's kernels.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline
  python scripts/knn_timeline.py /tmp/tl/t_kernel_trace.csv [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ks = [r for r in rows if any(t in r["Kernel_Name"] for t in ("knn_", "ratio_scatter"))]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: the timed region's dispatches are the last ones before the profiled (solo) launch sets; take a window well inside
mid = ks[len(ks) // 2 - n // 2: len(ks) // 2 + n // 2]
t0 = int(mid[0]["Start_Timestamp"])
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
for r in mid:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    ov = 0.0
    for o in mid:
        if o is r or o["Queue_Id"] == r["Queue_Id"]:
            continue
        os_, oe = (int(o["Start_Timestamp"]) - t0) / 1e3, (int(o["End_Timestamp"]) - t0) / 1e3
        ov += max(0.0, min(e, oe) - max(s, os_))
    print(f"q{r['Queue_Id']:>3s} {short(r['Kernel_Name']):28s} start {s:8.1f} end {e:8.1f} dur {e - s:6.1f} us  overlapped by other queue {ov:6.1f} us")
