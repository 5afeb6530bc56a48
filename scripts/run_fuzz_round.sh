#!/bin/bash
# The round's long KNN parity sweep on the FINAL binary (run via gpurun; logs land in gpurun_out/, copy them to profiles/).
# usage: bash scripts/run_fuzz_round.sh rNN [seconds per seed] [seed base: seeds base+1 .. base+5, default 100]
TAG=${1:-r03}; SEC=${2:-150}; BASE=${3:-100}
R=${GRAFT_REPO_ROOT:-/root/repo}
for seed in $((BASE+1)) $((BASE+2)) $((BASE+3)) $((BASE+4)); do
  python $R/scripts/fuzz_knn.py $SEC $seed 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/${TAG}_fuzz_knn_seed$seed.log
  tail -3 $R/gpurun_out/${TAG}_fuzz_knn_seed$seed.log | head -1
done
python $R/scripts/fuzz_knn.py $SEC $((BASE+5)) big 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/${TAG}_fuzz_knn_seed$((BASE+5))_big.log
python $R/scripts/fuzz_knn.py $((SEC*3/2)) $((BASE+6)) q8 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/${TAG}_fuzz_knn_seed$((BASE+6))_q8.log
grep "^fuzz:" $R/gpurun_out/${TAG}_fuzz_knn_seed*.log
# the other families (each log ends with the loaded binary's sfm_build_id: tests/test_gpu_knn.py checks every family against ITS sources)
OSEC=${4:-$SEC}
for seed in $((BASE+11)) $((BASE+12)); do
  python $R/scripts/fuzz_sift.py $OSEC $seed 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/${TAG}_fuzz_sift_seed$seed.log
  python $R/scripts/fuzz_geometry.py $OSEC $seed 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/${TAG}_fuzz_geometry_seed$seed.log
  python $R/scripts/fuzz_pipeline.py $OSEC $seed 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/${TAG}_fuzz_pipeline_seed$seed.log
done
grep -h "^fuzz_" $R/gpurun_out/${TAG}_fuzz_sift_seed*.log $R/gpurun_out/${TAG}_fuzz_geometry_seed*.log $R/gpurun_out/${TAG}_fuzz_pipeline_seed*.log | cut -c1-200
