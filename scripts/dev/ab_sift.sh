#!/bin/bash
# Dev: same-box A/B of the SIFT kernels of several library builds (kernel stats of `bench.py --workload sift`).
# usage (via gpurun): bash scripts/dev/ab_sift.sh "lib.so[:ENV=VAL]" ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for spec in "$@"; do
  L=${spec%%:*}; E=""; [ "$spec" != "$L" ] && E=${spec#*:}
  rm -rf /tmp/sst
  echo "== $spec (rep $rep)"
  env $E SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sst -o k -- python $R/bench.py --workload sift --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  frames/s %.0f  ms/frame %.3f  parity %s' % (d['value'], d['ms_per_step'], d.get('parity')))"
  python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/sst/k_kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if float(r['AverageNs']) * int(r['Calls']) > 2e5: print(f"  {n[:40]:40s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
done; done
