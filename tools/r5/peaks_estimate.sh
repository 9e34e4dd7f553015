#!/bin/bash
# one GPU session for SPX_OPT_PEAKS_ESTIMATE: the parity tests, then config2_peaks with the float64 passes and with the interval
# estimates in every tiling (sweep = one row per distinct cpu request; every_row = all 100 000 rows)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/peaks
timeout 300 python -m pytest tests/test_gpu_peaks.py -x -q 2>&1 | tail -15
for v in ${PEAKS_VARIANTS:-0 1 164 1616 84 88}; do
  timeout 120 python bench.py --workload config2_peaks --sweep-only --cpu-budget 0 --steps 10 --warmup 3 --opt PEAKS_ESTIMATE=$v > gpurun_out/peaks/line_$v.json 2> gpurun_out/peaks/err_$v.txt
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/peaks/line_{v}.json").read().strip().splitlines()[-1])
    er = d.get("every_row") or {}
    print("PEAKS_ESTIMATE", v, "sweep ms", round(d["ms_per_step"], 4), "every_row", {k: (round(x, 4) if isinstance(x, float) else x) for k, x in er.items() if "ms" in k})
except Exception as ex:
    print("PEAKS_ESTIMATE", v, "failed", ex, open(f"gpurun_out/peaks/err_{v}.txt").read()[-400:])
PY
done
