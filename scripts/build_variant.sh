#!/bin/bash
# Dev: build sfm_mvs_amd/lib/libsfmhip_<name>.so from the knn.hip of a git revision (other objects: the current build) for A/B runs
# usage: bash scripts/build_variant.sh <name> <git-rev>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
git -C $R show $2:sfm_mvs_amd/csrc/knn.hip > $R/sfm_mvs_amd/csrc/knn_variant_tmp.hip
( cd $R/sfm_mvs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -c knn_variant_tmp.hip -o $R/build/csrc/knn_$1.o; rm -f knn_variant_tmp.hip )
cd $R/build/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sfm_mvs_amd/lib/libsfmhip_$1.so assoc.o ba_dense.o ba_schur.o blocks.o common.o knn_$1.o ransac.o residual.o sift.o triangulate.o
ls -la $R/sfm_mvs_amd/lib/libsfmhip_$1.so
