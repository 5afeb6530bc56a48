#!/bin/bash
# Dev: kernel timeline of the pipelined bench step (3 launch sets in flight) on the GPU box
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl; mkdir -p /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras "$@" > /tmp/tl_bench.json 2>/dev/null
python $R/scripts/timeline.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1)
