#!/bin/bash
# per-kernel times of the NRT sweep kernels for library variants (tools/variant.py), same box, alternating
#   tools/r4/ab_nrt.sh "<workload> [<workload> ...]" <variant> [<variant> ...]
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
WL="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    for w in $WL; do
      out=gpurun_out/ab_nrt/$v.$w.$rep
      rm -rf $out && mkdir -p $out
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/variant.py run $v bench.py --workload $w --steps 20 --warmup 3 --cpu-budget 0 --sweep-only > $out/line.json 2> $out/err.log
      f=$(find $out -name '*kernel_stats.csv' | head -1)
      echo "== $v $w rep$rep ms_per_step=$(python -c "import json,sys; print(round(json.loads(open('$out/line.json').read().strip().splitlines()[-1])['ms_per_step'],4))" 2>/dev/null)"
      python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_nrt" in r["Name"] or "k_rows" in r["Name"]:
        print("   %-60s calls %4s avg %9.1f us" % (r["Name"].split("::")[-1][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    done
  done
done
