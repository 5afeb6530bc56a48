#!/bin/bash
# The multi-GPU scaling curve in one command, for the day an N-GPU node is available:
#   bash scripts/run_scale.sh [workload: knn | c5 | allpairs] [gpu counts, default "1 2 4 8"]
# Each count runs bench.py as the driver does (one rank per GPU over RCCL, rendezvous on 127.0.0.1) and appends the JSON line to
# gpurun_out/scale_<workload>.jsonl; the last line greps what the exchange really was (RCCL rank count / exchange kind).
WL=${1:-c5}; shift
COUNTS=${*:-1 2 4 8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/scale_$WL.jsonl; : > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in $COUNTS; do
  if [ "$N" = 1 ]; then
    python $R/bench.py --workload $WL --gpus 1 >> $OUT 2>> $R/gpurun_out/scale_$WL.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      $R/bench.py --workload $WL --gpus $N >> $OUT 2>> $R/gpurun_out/scale_$WL.err
  fi
done
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print(d["n_gpus"], "GPUs:", "%.4g" % d["value"], d["unit"], "| scaling", d.get("scaling"), "| exchange:", d.get("config", {}).get("exchange", d.get("config", {}).get("parallelism")))
PY
