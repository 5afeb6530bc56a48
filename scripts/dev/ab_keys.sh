R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for r in 1 2 3; do for L in libsfmhip_old.so libsfmhip.so; do echo "== $L rep $r"; SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L SFM_BATCH=8 bash $R/scripts/dev/kstats.sh 60 2>&1 | grep -E "refine|filter_q4|prep"; done; done
for L in libsfmhip_old.so libsfmhip.so; do echo "== U8 $L"; SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L SFM_BATCH=8 bash $R/scripts/dev/kstats.sh 60 10000 10000 sift 2>&1 | grep -E "refine|filter_q4"; done
for L in libsfmhip_old.so libsfmhip.so libsfmhip_old.so libsfmhip.so; do echo "== bench $L"; SFM_HIP_LIB=$R/sfm_mvs_amd/lib/$L python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
