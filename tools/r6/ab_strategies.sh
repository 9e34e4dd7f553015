#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
for wl in config3_most config3_balanced config3_leastnuma config3_r8 config3_r8_balanced; do
  st="--steps 20 --warmup 4"; [ $wl = config3_leastnuma ] && st="--steps 8 --warmup 2"
  python bench.py --workload $wl $st --sweep-only --cpu-budget 0 --no-every-row > gpurun_out/r6/st_${wl}.json 2> gpurun_out/r6/st_${wl}.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6/st_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "kernel_ms", round(d["roofline"]["kernel_ms"], 4))
    except Exception as ex:
        print(f, "ERR", ex)
PY
