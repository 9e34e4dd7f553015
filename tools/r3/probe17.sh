#!/bin/bash
export SPX_QOS_ONLY=1
for v in abl16 abl48; do SPX_VARIANT=$v timeout 200 python tools/r3/exp_qos.py LeastNUMANodes 2>&1 | tail -1; done
