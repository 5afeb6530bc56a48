#!/bin/bash
# Dev: A/B of library builds on ONE box, batched step.  usage: bash scripts/ab2.sh <reps> lib1.so lib2.so ...   (paths relative to the repo)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
reps=$1; shift
for r in $(seq 1 $reps); do
  for L in "$@"; do
    echo "== $L (rep $r)"
    SFM_HIP_LIB=$R/$L python $R/scripts/run_knn_batch.py 100 4 1 2>/dev/null | grep batch
    SFM_HIP_LIB=$R/$L python $R/scripts/run_knn_batch.py 300 4 3 2>/dev/null | grep batch
    SFM_HIP_LIB=$R/$L python $R/scripts/run_knn_batch.py 200 1 1 2>/dev/null | grep batch
  done
done
