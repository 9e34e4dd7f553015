mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_nrt.py tests/test_gpu_property.py -x -q -m gpu > gpurun_out/r6/t_nrt.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r6/t_nrt.log | tail -5
