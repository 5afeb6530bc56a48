#!/bin/bash
# Dev: same-box A/B of two library builds: batched kernel stats (uniform = quantised body, SIFT-like = exact-integer body) + bench line.
# usage (via gpurun): bash scripts/ab_libs.sh libsfmhip_old.so libsfmhip.so
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
stats() {   # $1 lib, rest: run_knn_* command
  L=$1; shift
  rm -rf /tmp/kst
  SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L SFM_BATCH=8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- "$@" 2>/dev/null | grep done
  python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/kst/k_kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if n.startswith('knn_') or n.startswith('ratio_'): print(f"  {n[:44]:44s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
}
for rep in 1 2; do for L in "$@"; do
  echo "== uniform $L (rep $rep)"; stats $L python $R/scripts/run_knn_steps.py 60
  echo "== sift-like $L (rep $rep)"; SFM_WARM=20 stats $L python $R/scripts/run_knn_sift.py 60 8 1
done; done
for rep in 1 2; do for L in "$@"; do
  echo "== bench $L"; SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('sift_like',{}); print('value %.4g ms/step %.4f cold %.4g | sift-like %.4g ms/step %.4f | kernels' % (d['value'], d['ms_per_step'], d['cold_value'], s.get('distances_per_sec',0), s.get('ms_per_step',0)), d['kernels_ms'], s.get('kernels_ms'))"
done; done
