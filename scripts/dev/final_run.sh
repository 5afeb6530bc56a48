cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r05; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
S=$(date +%s); timeout 900 python bench.py > $O/r05_bench_default.json 2> $O/bench_default.err; echo "default bench wall $(( $(date +%s) - S )) s"; python - <<'PY'
import json
d = json.load(open("gpurun_out/final_r05/r05_bench_default.json"))
print({k: d[k] for k in ("metric", "value", "ms_per_step", "n_gpus")}, d["roofline"], d["cpu_baseline"], d["config"].get("secondary"))
PY
timeout 300 python bench.py --workload sift --steps 30 --warmup 5 > $O/r05_bench_sift.json 2>> $O/bench_default.err; python -c "
import json; d = json.load(open('$O/r05_bench_sift.json')); print('sift', d['value'], d['ms_per_step'], d['frame_latency_ms_single_stream'], d['descriptor_kernel'], d.get('parity'))"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/t1 -o sift1 -- python $R/bench.py --workload sift --steps 30 --warmup 5 --pipe-depth 1 --no-cpu-baseline > /dev/null 2>&1; cp $R/$O/t1/sift1_kernel_stats.csv $R/$O/r05_sift_depth1_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/t3 -o sift -- python $R/bench.py --workload sift --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; cp $R/$O/t3/sift_kernel_stats.csv $R/$O/r05_sift_kernel_stats.csv
rm -rf $R/$O/t1 $R/$O/t3
