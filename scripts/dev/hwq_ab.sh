#!/bin/bash
# Dev: the default bench run under different GPU_MAX_HW_QUEUES (HIP maps streams onto that many hardware queues, round robin).
cd $GRAFT_REPO_ROOT
for q in default 8 default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['config']['secondary']
print('hw queues $q: value %.4g ms/step %.4f | sift-like %.4g | sift %.0f f/s | sfm57 %.4f s | from pixels %.4f s | c5 %.4g | allpairs %.4g' % (d['value'], d['ms_per_step'], s['sift_like_u8_distances_per_sec'], s['sift_frames_per_sec'], s['sfm57_from_features_seconds'], s['sfm57_from_pixels_seconds'], s['config5_one_gpu_distances_per_sec'], s['allpairs_distances_per_sec']))"
done
