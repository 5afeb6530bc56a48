#!/bin/bash
# Dev: per-kernel averages (rocprofv3 --stats) of one batched KNN run for several library builds on ONE box.
# usage: bash scripts/dev/kstats_ab.sh lib1 lib2 ...   (names under sfm_mvs_amd/lib, without .so)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$L -o k -- python $R/scripts/run_knn_batch.py 100 8 1 > /dev/null 2>&1
  python - /tmp/p_$L/k_kernel_stats.csv $L <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for k in ("knn_prep", "knn_split_images", "knn_filter_q4", "knn_refine", "ratio_scatter"):
        if k in n:
            out.append(f"{k} {float(r['AverageNs']) / 1e3:.1f}")
print(sys.argv[2], "|", " | ".join(sorted(out)))
PY
done
