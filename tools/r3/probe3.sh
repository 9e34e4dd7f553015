#!/bin/bash
export SPX_NRT_CPB=1
for v in "" f5l5 f4l3 f3l3; do echo "variant=$v"; SPX_VARIANT=$v timeout 200 python tools/r3/exp_qos.py LeastAllocated 2>&1 | tail -1; done
