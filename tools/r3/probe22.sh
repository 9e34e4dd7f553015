#!/bin/bash
for rep in 1 2; do
for v in "" insweep; do
  SPX_VARIANT=$v timeout 120 python tools/r3/bench_variant.py --workload config2 --sweep-only --cpu-budget 0 --steps 200 --warmup 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=$v', round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"
done; done
timeout 200 python -m pytest tests/test_gpu_trimaran.py -m gpu -x -q 2>&1 | tail -2
