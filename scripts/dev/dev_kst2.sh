cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 0 1; do
  if [ $e = 1 ]; then export SFM_BENCH_EXCHANGE=1; else unset SFM_BENCH_EXCHANGE; fi
  rm -rf /tmp/kx$e
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx$e -o k -- python $R/bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== exchange=$e"
  python - <<PY
import csv
for r in list(csv.DictReader(open('/tmp/kx$e/k_kernel_stats.csv')))[:9]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"  {n[:60]:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
done
