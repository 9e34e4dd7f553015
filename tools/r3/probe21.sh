#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/commit; rm -rf $O; mkdir -p $O
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $R/tools/prof_commit.py 3000 direct > $O/t.log 2>&1
tail -1 $O/t.log
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"])>0.3: print(f'   {r["Name"][:86]:86s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]}%')
PY
find $O -name "*kernel_trace.csv" -delete
