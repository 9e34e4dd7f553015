#!/bin/bash
export SPX_QOS_ONLY=1
timeout 200 python tools/r3/exp_qos.py MostAllocated 2>&1 | tail -1
SPX_VARIANT=m3 timeout 200 python tools/r3/exp_qos.py MostAllocated 2>&1 | tail -1
mkdir -p gpurun_out/r3
timeout 200 python bench.py --workload small_full --devices 0,0 --transport copy --gather table --steps 5 --warmup 2 --cpu-budget 0 2>/dev/null | tail -1 > gpurun_out/r3/multi_small_full.json
head -c 600 gpurun_out/r3/multi_small_full.json
