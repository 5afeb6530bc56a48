#!/bin/bash
# Dev: A/B of an environment switch on one box, batch of 8 under rocprofv3.  usage: bash scripts/abl_env.sh VAR v1 v2 ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
V=$1; shift
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for a in "$@"; do
  rm -rf /tmp/kst
  env $V=$a SFM_BATCH=8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/scripts/run_knn_steps.py 100 2>/dev/null | grep done | cut -c1-90
  python - "$V=$a" <<'PY'
import csv, sys
for r in csv.DictReader(open('/tmp/kst/k_kernel_stats.csv')):
    if 'knn_filter' in r['Name']: print(f"{sys.argv[1]}  filter avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  calls {r['Calls']}")
PY
done
done
