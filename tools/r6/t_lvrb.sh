#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_trimaran.py tests/test_gpu_property.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_exhaustive.py -x -q -m gpu -k "config2" 2>&1 | tail -3
for wl in config2_lvrb; do
  python bench.py --workload $wl --steps 100 --warmup 20 --sweep-only --cpu-budget 0 --no-every-row > gpurun_out/r6/lv_${wl}.json 2> gpurun_out/r6/lv_${wl}.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r6/lv_${wl}.json").read().strip().splitlines()[-1])
print("${wl}", "ms_per_step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4))
PY
done
bash tools/r6/trace1.sh config2_lvrb med3
