#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/fault; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_exhaustive.py -m gpu -x -q -k full_cycle_every_row 2>&1 | tail -3
cd /tmp
for i in 1 2; do
  AMD_SERIALIZE_KERNEL=3 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$i -o t -- python $R/bench.py --workload config2 --cpu-budget 0 > $O/t$i.log 2>&1
  echo "serialized run $i rc=$?"; tail -2 $O/t$i.log | cut -c1-300
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t3 -o t -- python $R/bench.py --workload config2 --cpu-budget 0 > $O/t3.log 2>&1
echo "plain profiled run rc=$?"; tail -1 $O/t3.log | cut -c1-200
find $O -name "*kernel_trace.csv" -delete
