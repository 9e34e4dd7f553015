#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_nrt.py -m gpu -x -q 2>&1 | tail -2
for c in 1 2 4 8 16; do echo "CPB=$c"; SPX_NRT_CPB=$c timeout 200 python tools/r3/exp_qos.py LeastAllocated 2>&1 | tail -1; done
