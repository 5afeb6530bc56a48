#!/bin/bash
# The multi-GPU scaling curve in one command, for the day an N-GPU node is available:
#   bash scripts/run_scale.sh [workload: knn | c5 | allpairs] [gpu counts, default "1 2 4 8"]
# bench.py --gpus N starts its own N ranks (one per GPU over RCCL, rendezvous on 127.0.0.1: bench.py self_launch), exactly as the
# driver's command line does; each JSON line is appended to gpurun_out/scale_<workload>.jsonl.
WL=${1:-c5}; shift
COUNTS=${*:-1 2 4 8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/scale_$WL.jsonl; : > $OUT
for N in $COUNTS; do
  python $R/bench.py --workload $WL --gpus $N >> $OUT 2>> $R/gpurun_out/scale_$WL.err
done
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print(d["n_gpus"], "GPUs:", "%.4g" % d["value"], d["unit"], "| scaling", d.get("scaling"), "| exchange:", d.get("config", {}).get("exchange", d.get("config", {}).get("parallelism")))
PY
