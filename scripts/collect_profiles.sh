#!/bin/bash
# Round profile collection on the GPU box: kernel-trace stats of bench.py + PMC passes of the KNN step.
# Usage (via gpurun): bash scripts/collect_profiles.sh rNN      → files under gpurun_out/profiles_rNN/
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o knn -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
cp $OUT/trace/knn_kernel_stats.csv $OUT/${TAG}_knn_kernel_stats.csv
# the same bench with ONE pair in flight: kernel durations without the neighbouring pairs' kernels sharing the chip
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o knn1 -- python $R/bench.py --steps 100 --warmup 10 --pipe-depth 1 --no-cpu-baseline --no-extras > $OUT/bench_depth1_under_rocprof.json 2>> $OUT/trace.log
cp $OUT/trace1/knn1_kernel_stats.csv $OUT/${TAG}_knn_depth1_kernel_stats.csv
# one launch set per step and nothing else (bench.py also launches single pairs for its latency and variant legs, which
# mix into the averages above): the per-kernel durations of the batch the roofline is quoted on
SFM_BATCH=${SFM_PROFILE_BATCH:-8} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/traceb -o knnb -- python $R/scripts/run_knn_steps.py 60 > $OUT/knn_batch_steps.log 2>> $OUT/trace.log
cp $OUT/traceb/knnb_kernel_stats.csv $OUT/${TAG}_knn_batch_kernel_stats.csv
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
P2="FETCH_SIZE"
P3="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES"
i=1
BATCH=${SFM_PROFILE_BATCH:-8}          # pairs per launch set: what bench.py runs by default
for P in "$P1" "$P2" "$P3" "$P4"; do
  SFM_BATCH=$BATCH rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc -o pass$i -- python $R/scripts/run_knn_steps.py 6 > $OUT/pmc_pass$i.log 2>&1
  i=$((i+1))
done
python $R/scripts/summarize_pmc.py $OUT/pmc $TAG $BATCH > $OUT/${TAG}_knn_pmc.md
# ... and on the same uniform data through filter = noquant: the fp16 body the headline ran before the quantised integer body
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  SFM_FILTER=noquant SFM_BATCH=$BATCH rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_f16 -o pass$i -- python $R/scripts/run_knn_steps.py 6 > $OUT/pmc_f16_pass$i.log 2>&1
  i=$((i+1))
done
python $R/scripts/summarize_pmc.py $OUT/pmc_f16 $TAG $BATCH f16 > $OUT/${TAG}_knn_f16_pmc.md
SFM_FILTER=noquant SFM_BATCH=$BATCH rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tracef -o knnf -- python $R/scripts/run_knn_steps.py 60 > $OUT/knn_f16_batch_steps.log 2>> $OUT/trace.log
cp $OUT/tracef/knnf_kernel_stats.csv $OUT/${TAG}_knn_f16_batch_kernel_stats.csv
rm -rf $OUT/tracef
# the same four passes on SIFT-like u8 descriptors: the exact-integer (i8 MFMA) body and its refine
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  SFM_WARM=2 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_i8 -o pass$i -- python $R/scripts/run_knn_sift.py 6 $BATCH 1 > $OUT/pmc_i8_pass$i.log 2>&1
  i=$((i+1))
done
python $R/scripts/summarize_pmc.py $OUT/pmc_i8 $TAG $BATCH i8 > $OUT/${TAG}_knn_i8_pmc.md
SFM_WARM=20 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_i8 -o knni8 -- python $R/scripts/run_knn_sift.py 60 $BATCH 1 > $OUT/knn_i8_steps.log 2>> $OUT/trace.log
cp $OUT/trace_i8/knni8_kernel_stats.csv $OUT/${TAG}_knn_i8_batch_kernel_stats.csv
for wl in tri ba; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -o $wl -- python $R/bench.py --workload $wl --steps 5 --warmup 1 > $OUT/bench_${wl}_under_rocprof.json 2>> $OUT/trace.log
  cp $OUT/trace_$wl/${wl}_kernel_stats.csv $OUT/${TAG}_${wl}_kernel_stats.csv
done
# SIFT leg: one frame in flight (kernel durations as they are alone) and the default three
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sift1 -o sift1 -- python $R/bench.py --workload sift --steps 30 --warmup 5 --pipe-depth 1 --no-cpu-baseline > $OUT/bench_sift_depth1_under_rocprof.json 2>> $OUT/trace.log
cp $OUT/trace_sift1/sift1_kernel_stats.csv $OUT/${TAG}_sift_depth1_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sift -o sift -- python $R/bench.py --workload sift --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_sift_under_rocprof.json 2>> $OUT/trace.log
cp $OUT/trace_sift/sift_kernel_stats.csv $OUT/${TAG}_sift_kernel_stats.csv
rm -rf $OUT/trace_i8 $OUT/trace $OUT/traceb $OUT/trace1 $OUT/trace_tri $OUT/trace_ba $OUT/trace_sift $OUT/trace_sift1
# the bench lines themselves, un-profiled (a profiled run clocks 2-5 % lower): the driver's flags, config 5 through a
# one-rank RCCL group, the 57-camera driver
cd $R
# (round 6: stdout carries the compact line; the FULL record of a run is gpurun_out/bench_full.json — that is what is kept as rNN_bench_*.json,
#  the line itself beside it as *.line)
runb() { name=$1; shift; python bench.py "$@" > $OUT/${TAG}_bench_$name.line 2>> $OUT/trace.log; cp $R/gpurun_out/bench_full.json $OUT/${TAG}_bench_$name.json; }
runb default --gpus 1 --steps 20 --warmup 5
cp $OUT/${TAG}_bench_default.json $OUT/${TAG}_bench_knn.json
runb c5 --workload c5
runb allpairs --workload allpairs
runb sfm --workload sfm --steps 3
runb sfm_pixels --workload sfm --from-pixels --steps 3
# launch sets in flight: the headline step at pipeline depth 1 .. 5 (bench.py's default is 2)
for d in 1 2 3 4 5; do
  python bench.py --steps 60 --warmup 10 --pipe-depth $d --no-cpu-baseline --no-extras 2>> $OUT/trace.log | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('pipe-depth $d: value %.4g distances/s  ms_per_step %.4f  frac_step %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac_step']))" >> $OUT/${TAG}_knn_pipe_depth.txt
done
# HIP API census of the from-pixels job (how often the host waits: hipStreamSynchronize / hipMemcpy / hipEventSynchronize)
( cd /tmp && rocprofv3 --hip-trace --stats --output-format csv -d $OUT/trace_hip -o sfmpx -- python $R/bench.py --workload sfm --from-pixels --steps 1 --no-cpu-baseline > /dev/null 2>> $OUT/trace.log )
cp $OUT/trace_hip/sfmpx_hip_api_stats.csv $OUT/${TAG}_sfm_pixels_hip_api_stats.csv 2>/dev/null
rm -rf $OUT/trace_hip
runb tri --workload tri --steps 5 --warmup 1
runb ba --workload ba --steps 5 --warmup 1
runb sift --workload sift --steps 30 --warmup 5
# PMC passes of the non-KNN legs (triangulation, BA, SIFT)
bash $R/scripts/collect_pmc_other.sh $TAG > $OUT/pmc_other.log 2>&1
python $R/scripts/summarize_pmc_other.py $TAG > $OUT/${TAG}_other_pmc.md
ls $OUT
