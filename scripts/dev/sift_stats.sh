#!/bin/bash
# Per-kernel breakdown of SIFT under rocprofv3 (run on the GPU box via gpurun).  Usage: bash scripts/sift_stats.sh [steps] [w] [h]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sst -o k -- python $R/scripts/run_sift_steps.py "${@:-20}" 2>/dev/null | grep done
python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/sst/k_kernel_stats.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"  {n[:44]:44s} calls {r['Calls']:>5s}  total {float(r['TotalDurationNs'])/1e3:9.1f} us  avg {float(r['AverageNs'])/1e3:8.2f} us  max {float(r['MaxNs'])/1e3:8.2f}  {r['Percentage']}%")
PY
mkdir -p $R/gpurun_out && cp /tmp/sst/k_kernel_stats.csv $R/gpurun_out/sift_kernel_stats.csv
python - <<'PY'
import csv
rows = [r for r in csv.DictReader(open('/tmp/sst/k_kernel_trace.csv'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last frame: from the last upsample2 launch on
last = max(i for i, r in enumerate(rows) if 'upsample2' in r['Kernel_Name'])
t0 = int(rows[last]['Start_Timestamp'])
print("  -- last frame timeline (us from upsample start: start, duration, kernel, grid)")
for r in rows[last:]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {n[:28]:28s} {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
PY
