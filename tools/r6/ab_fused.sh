#!/bin/bash
# A/B of the fused NRT sweep on one box: config3 and config5_share with SPX_OPT_NRT_FUSED 1 / 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
for wl in config3 config5_share; do
  for f in 1 0; do
    python bench.py --workload $wl --steps 30 --warmup 5 --sweep-only --cpu-budget 0 --opt NRT_FUSED=$f > gpurun_out/r6/ab_${wl}_fused$f.json 2> gpurun_out/r6/ab_${wl}_fused$f.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "every_row", round(d.get("every_row", {}).get("kernel_ms", 0), 4))
    except Exception as ex:
        print(f, "ERR", ex)
PY
