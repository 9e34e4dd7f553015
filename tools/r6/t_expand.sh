#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_nrt.py tests/test_gpu_peaks.py -x -q -m gpu 2>&1 | tail -3
for wl in config2_peaks config3 config5_share; do
  python bench.py --workload $wl --steps 20 --warmup 4 --sweep-only --cpu-budget 0 --no-every-row > gpurun_out/r6/ex_${wl}.json 2> gpurun_out/r6/ex_${wl}.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r6/ex_${wl}.json").read().strip().splitlines()[-1])
print("${wl}", "ms_per_step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4))
PY
done
