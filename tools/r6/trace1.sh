#!/bin/bash
# kernel-trace stats of one workload's sweep: tools/r6/trace1.sh <workload> <tag> [bench options]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; TAG=$2; shift 2
OUT=$R/gpurun_out/r6/trace_${W}_$TAG
mkdir -p $OUT
cd /tmp
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 20 --warmup 3 $* > $OUT/trace.log 2>&1
rm -f $OUT/trace/*/t_kernel_trace.csv $OUT/trace/t_kernel_trace.csv
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs head -9 | cut -c1-200 | grep -v vectorized
