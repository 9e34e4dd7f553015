#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_peaks.py tests/test_gpu_property.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_exhaustive.py -x -q -m gpu -k "peaks" 2>&1 | tail -3
