#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/kt; rm -rf $O; mkdir -p $O
cd /tmp
for S in LeastAllocated MostAllocated BalancedAllocation LeastNUMANodes; do
  timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$S -o t -- python $R/tools/r3/exp_one.py $S 0.5,0.4,0.1 5 > $O/$S.log 2>&1
  echo "== $S"; tail -1 $O/$S.log
  f=$(find $O/$S -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"])>0.5: print(f'   {r["Name"][:80]:80s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:.3f} ms  {r["Percentage"]}%')
PY
  find $O/$S -name "*kernel_trace.csv" -delete
done
