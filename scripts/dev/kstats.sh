#!/bin/bash
# Per-kernel breakdown of the KNN step under rocprofv3 (run on the GPU box via gpurun).
# Usage: bash scripts/kstats.sh [steps] [nq] [nt] [uniform|sift]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/scripts/run_knn_steps.py "${@:-100}" 2>/dev/null | grep done
python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/kst/k_kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"  {n[:44]:44s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
PY
