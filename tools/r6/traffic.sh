#!/bin/bash
# kernel stats + WRITE_SIZE / FETCH_SIZE passes of one workload: tools/r6/traffic.sh <workload> <tag>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; TAG=$2; shift 2
OUT=$R/gpurun_out/r6/tr_${W}_$TAG
mkdir -p $OUT
cd /tmp
B="python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 3 --warmup 1 $*"
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 20 --warmup 5 $* > $OUT/trace.log 2>&1
timeout 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -o p -- $B > $OUT/pmc3.log 2>&1
timeout 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc4 -o p -- $B > $OUT/pmc4.log 2>&1
rm -f $OUT/trace/*/t_kernel_trace.csv $OUT/trace/t_kernel_trace.csv
python $R/tools/pmc_summary.py $(find $OUT/pmc3 $OUT/pmc4 -name "*counter_collection.csv") 2>/dev/null | grep -A2 "spx::" | grep -v "^--" | head -60
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-160
