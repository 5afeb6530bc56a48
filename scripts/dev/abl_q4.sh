#!/bin/bash
# Dev: timing ablations of the q4 filter (SFM_KNN_ABL values given as arguments) on one box, batch of 8 under rocprofv3
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for a in "$@"; do
  rm -rf /tmp/kst
  SFM_KNN_ABL=$a SFM_BATCH=8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/scripts/run_knn_steps.py 60 >/dev/null 2>&1
  python - "$a" <<'PY'
import csv, sys
for r in csv.DictReader(open('/tmp/kst/k_kernel_stats.csv')):
    if 'knn_filter' in r['Name']: print(f"ABL={sys.argv[1]}  filter avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  calls {r['Calls']}")
PY
done
done
