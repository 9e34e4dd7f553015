#!/bin/bash
# round 3, probe 1: baseline of the NRT workloads on this build + first execution of bench.py's multi-device mode
mkdir -p gpurun_out/r3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
for w in config3 config3_most config3_balanced config3_leastnuma; do
  timeout 300 python bench.py --workload $w --sweep-only --cpu-budget 0 --steps 20 --warmup 5 > gpurun_out/r3/p1_$w.json 2> gpurun_out/r3/p1_$w.err
done
timeout 300 python bench.py --workload config5_share --sweep-only --cpu-budget 0 --steps 10 --warmup 3 > gpurun_out/r3/p1_c5.json 2> gpurun_out/r3/p1_c5.err
timeout 300 python bench.py --workload small --devices 0,0 --transport copy --steps 3 --warmup 1 --cpu-budget 0 > gpurun_out/r3/p1_multi_small.json 2> gpurun_out/r3/p1_multi_small.err
timeout 300 python bench.py --workload small --devices 0,0 --transport copy --steps 3 --warmup 1 --cpu-budget 0 --gather table > gpurun_out/r3/p1_multi_small_table.json 2> gpurun_out/r3/p1_multi_small_table.err
timeout 300 python bench.py --workload ref_net_1000 --devices 0,0 --transport copy --steps 3 --warmup 1 --cpu-budget 0 --gather table > gpurun_out/r3/p1_multi_net.json 2> gpurun_out/r3/p1_multi_net.err
for f in gpurun_out/r3/p1_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("n_gpus","ms_per_step","value")}, d["roofline"]["kernel_ms"], d.get("gather"), d.get("topological_sort"))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
