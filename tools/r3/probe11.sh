#!/bin/bash
timeout 400 python -m pytest tests/test_gpu_nrt.py -m gpu -x -q 2>&1 | tail -2
SPX_NOSIDE=1 timeout 200 python tools/r3/exp_qos.py LeastAllocated MostAllocated BalancedAllocation 2>&1 | tail -1
timeout 200 python tools/r3/exp_qos.py LeastAllocated MostAllocated BalancedAllocation 2>&1 | tail -1
timeout 300 python bench.py --workload config5_share --sweep-only --cpu-budget 0 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', round(d['roofline']['kernel_ms'],3))"
