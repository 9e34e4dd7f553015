#!/bin/bash
# round 6, final build: every workload whose kernels changed since its profile (stamps: tests/test_profile_stamps.py), one gpurun call per batch
set -u
cd /root/repo
for batch in "config2 config2_lvrb config2_peaks" "config3 config3_most config3_balanced" "config3_leastnuma config3_r8 config3_r8_balanced" "config5_share"; do
  rm -rf gpurun_out/prof_*
  /usr/local/graft/bin/gpurun --timeout 1500 -- "bash tools/prof_all.sh $batch 2>&1 | tail -4" 2>&1 | grep -E "gpurun\]|kernel_ms" | tail -4
  python tools/collect_profiles.py r06 $batch 2>&1 | tail -4
done
rm -rf gpurun_out/prof_*
/usr/local/graft/bin/gpurun --timeout 2400 -- 'mkdir -p gpurun_out/r6; timeout 1800 python bench.py --workload config5 --gpus 1 --steps 3 --warmup 1 --sweep-only --cpu-budget 0 > gpurun_out/r6/config5_full_bench_line.json 2> gpurun_out/r6/config5_full.err; python -c "
import json;d=json.loads(open(\"gpurun_out/r6/config5_full_bench_line.json\").read().strip().splitlines()[-1]);print(\"config5 whole\", d[\"roofline\"][\"kernel_ms\"], d.get(\"every_row\",{}).get(\"kernel_ms\"))"' 2>&1 | tail -2
cp gpurun_out/r6/config5_full_bench_line.json profiles/r06/
