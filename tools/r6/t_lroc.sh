mkdir -p gpurun_out/r6
python -m pytest tests -x -q -m gpu -k "lroc or LROC" > gpurun_out/r6/t_lroc.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r6/t_lroc.log | tail -5
