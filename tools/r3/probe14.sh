#!/bin/bash
timeout 400 python -m pytest tests/test_gpu_nrt.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python tools/r3/exp_qos.py LeastAllocated BalancedAllocation LeastNUMANodes 2>&1 | tail -1
