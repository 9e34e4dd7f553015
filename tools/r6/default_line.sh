#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
python bench.py > gpurun_out/r6/default_line.json 2> gpurun_out/r6/default_line.err
tail -c 300 gpurun_out/r6/default_line.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/default_line.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","ms_per_step","n_gpus","steps","warmup","dtype","vs_baseline")})
print("roofline", d["roofline"])
for k in ("config3_leg","config4_leg","config5_share_leg"):
    print(k, {kk:vv for kk,vv in d[k].items() if kk in ("kernel_ms","frac","every_row","algorithmic_bytes")})
print("cpu_baseline", d["cpu_baseline"])
print("full_cycle", d.get("full_cycle"))
print("config5_leg", d.get("config5_leg"))
PY
