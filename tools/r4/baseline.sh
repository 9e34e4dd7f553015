#!/bin/bash
# one lease: the -m gpu suite without the every-cell tests, then the sweep-only bench line of every workload -> gpurun_out/base/
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
mkdir -p gpurun_out/base
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_exhaustive.py -p no:cacheprovider > gpurun_out/base/pytest.log 2>&1
tail -3 gpurun_out/base/pytest.log
for w in "$@"; do
  timeout 120 python bench.py --workload $w --sweep-only --cpu-budget 0 > gpurun_out/base/$w.json 2> gpurun_out/base/$w.err
  python - $w <<'PY'
import json, sys
w = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/base/{w}.json").read().strip().splitlines()[-1])
    print(w, "ms_per_step", round(d["ms_per_step"], 4), "kernel_ms", d["roofline"].get("kernel_ms"), "frac", round(d["roofline"]["frac"], 4))
except Exception as e:
    print(w, "FAILED", e)
PY
done
