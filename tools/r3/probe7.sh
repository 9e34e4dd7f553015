#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_nrt.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python tools/r3/exp_qos.py LeastAllocated MostAllocated 2>&1 | tail -1
