#!/bin/bash
# where the time of the interval-estimate Peaks passes goes: per-kernel times of the every-row sweep, then the same sweep with the float64
# evaluations compiled out of the min/max pass (pkd1), the write pass (pkd2), both (pkd3) — tools/variant.py builds, wrong tables
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out/peaks
B="--workload config2_peaks --sweep-only --cpu-budget 0 --steps 10 --warmup 3 --no-pod-classes --no-every-row"
for v in ${PEAKS_VARIANTS:-1}; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/peaks/trace_$v -o t -- python $R/bench.py $B --opt PEAKS_ESTIMATE=$v > $R/gpurun_out/peaks/trace_$v.log 2>&1)
  f=$(find gpurun_out/peaks/trace_$v -name "*kernel_stats.csv" | head -1)
  echo "== PEAKS_ESTIMATE=$v kernel stats"; [ -n "$f" ] && cut -d, -f1-4 "$f" | head -8
done
for name in ${PEAKS_LIBS:-base pkd1 pkd2 pkd3}; do
  timeout 120 python tools/variant.py run $name bench.py $B > gpurun_out/peaks/diag_$name.json 2> gpurun_out/peaks/diag_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/peaks/diag_{n}.json").read().strip().splitlines()[-1])
    print("lib", n, "every-row sweep ms", round(d["ms_per_step"], 4))
except Exception as ex:
    print("lib", n, "failed", ex, open(f"gpurun_out/peaks/diag_{n}.err").read()[-300:])
PY
done
