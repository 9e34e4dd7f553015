#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_nrt.py -m gpu -x -q 2>&1 | tail -3
timeout 200 python tools/r3/exp_qos.py LeastAllocated 2>&1 | tail -1
timeout 200 python tools/r3/exp_qos.py MostAllocated BalancedAllocation LeastNUMANodes 2>&1 | tail -1
SPX_VARIANT=m3b3 timeout 200 python tools/r3/exp_qos.py MostAllocated BalancedAllocation 2>&1 | tail -1
