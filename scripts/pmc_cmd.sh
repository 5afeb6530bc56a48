#!/bin/bash
# Dev: PMC passes (separate rocprofv3 runs, counters only with --kernel-trace) around an arbitrary command.
# usage: bash scripts/pmc_cmd.sh <outdir-under-gpurun_out> "<counter group 1>" "<counter group 2>" ... -- <command ...>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
i=1
for P in "${GROUPS_[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -o pass$i -- "$@" > $OUT/pass$i.log 2>&1
  i=$((i+1))
done
python - "$OUT" <<'PY'
import collections, csv, glob, os, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pass*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if k.startswith("at::") or k.startswith("__amd") or k.startswith("Cijk"):
        continue
    print(f"## {k}  ({max(len(x) for x in v.values())} launches)")
    for c in sorted(v):
        print(f"   {c:36s} {sum(v[c]) / len(v[c]):16.6g}")
PY
