# Dev: same-box A/B of two library builds (kernel stats of the batched KNN step on uniform / SIFT-like data, then the bench line).
# Build the other library first, e.g. the previous commit's knn.hip:
#   git show HEAD~1:sfm_mvs_amd/csrc/knn.hip > build/ab/knn_old.hip && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
#     -Isfm_mvs_amd/csrc -c build/ab/knn_old.hip -o build/ab/knn_old.o && hipcc --offload-arch=gfx950 -shared -fPIC \
#     -o sfm_mvs_amd/lib/libsfmhip_old.so build/csrc/{assoc,ba_dense,ba_schur,blocks,common,ransac,residual,sift,triangulate}.o build/ab/knn_old.o
# then: gpurun -- 'bash scripts/dev/ab_keys.sh'   (SFM_HIP_LIB selects the library: sfm_mvs_amd/_lib.py)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for r in 1 2 3; do for L in libsfmhip_old.so libsfmhip.so; do echo "== $L rep $r"; SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L SFM_BATCH=8 bash $R/scripts/dev/kstats.sh 60 2>&1 | grep -E "refine|filter_q4|prep"; done; done
for L in libsfmhip_old.so libsfmhip.so; do echo "== U8 $L"; SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L SFM_BATCH=8 bash $R/scripts/dev/kstats.sh 60 10000 10000 sift 2>&1 | grep -E "refine|filter_q4"; done
for L in libsfmhip_old.so libsfmhip.so libsfmhip_old.so libsfmhip.so; do echo "== bench $L"; SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
