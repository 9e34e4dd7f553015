#!/bin/bash
export SPX_NRT_CPB=1
for v in "" abl1 abl2 abl4 abl8 abl5 abl7 abl15; do SPX_VARIANT=$v timeout 200 python tools/r3/exp_be.py 2>&1 | tail -2; done
