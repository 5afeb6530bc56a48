#!/bin/bash
# Dev: A/B of library builds on ONE box.  usage: bash scripts/ab.sh <reps> lib1.so lib2.so ...   (paths relative to the repo)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
reps=$1; shift
for r in $(seq 1 $reps); do
  for L in "$@"; do
    echo "== $L (rep $r)"
    SFM_HIP_LIB=$R/$L python $R/scripts/run_knn_steps.py 200 2>/dev/null | grep done
    SFM_HIP_LIB=$R/$L SFM_STREAMS=3 python $R/scripts/run_knn_steps.py 600 2>/dev/null | grep done
  done
done
