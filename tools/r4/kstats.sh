#!/bin/bash
# per-kernel rocprofv3 averages of the sweep kernels of bench workloads, for library variants (tools/variant.py), one lease
#   tools/r4/kstats.sh "<workload> [<workload> ...]" <variant> [<variant> ...]      (variant `base` = the in-tree library)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
WL="$1"; shift
for v in "$@"; do
  for w in $WL; do
    out=gpurun_out/kstats/$v.$w
    rm -rf $out && mkdir -p $out
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/variant.py run $v bench.py --workload $w --steps 20 --warmup 3 --cpu-budget 0 --sweep-only $EXTRA > $out/line.json 2> $out/err.log
    f=$(find $out -name '*kernel_stats.csv' | head -1)
    echo "== $v $w ms_per_step=$(python -c "import json,sys; print(round(json.loads(open('$out/line.json').read().strip().splitlines()[-1])['ms_per_step'],4))" 2>/dev/null)"
    python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 20 and float(r["AverageNs"]) > 3000:
        print("   %-72s calls %4s avg %9.1f us" % (r["Name"].replace("spx::(anonymous namespace)::", "").replace("void ", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    find $out -name '*kernel_trace.csv' -delete
  done
done
