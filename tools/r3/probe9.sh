#!/bin/bash
SPX_SINGLE=1 timeout 200 python tools/r3/exp_qos.py LeastAllocated 2>&1 | tail -1
SPX_SINGLE=1 SPX_VARIANT=both2 timeout 200 python tools/r3/exp_qos.py LeastAllocated 2>&1 | tail -1
