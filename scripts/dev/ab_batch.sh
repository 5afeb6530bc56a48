#!/bin/bash
# Dev: A/B of library builds on ONE box, batched launch sets under rocprofv3.  usage: bash scripts/ab_batch.sh <reps> lib1.so lib2.so ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
reps=$1; shift
cd /tmp && export TMPDIR=/tmp
for r in $(seq 1 $reps); do
  for L in "$@"; do
    echo "== $L (rep $r)"
    rm -rf /tmp/kst
    SFM_HIP_LIB=$R/$L SFM_BATCH=8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/scripts/run_knn_steps.py 60 2>/dev/null | grep done
    python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/kst/k_kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if n.startswith('knn_'): print(f"  {n[:44]:44s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
  done
done
