#!/bin/bash
# one gpurun call per batch of workloads (gpurun copies back at most 64 MiB), the judged summaries into profiles/r06 after each
set -u
cd /root/repo
for batch in "config2 config2_lvrb config2_lroc" "config2_peaks config3 config3_most" "config3_balanced config3_leastnuma config3_r8" "config3_r8_balanced config4 config5_share"; do
  rm -rf gpurun_out/prof_*
  /usr/local/graft/bin/gpurun --timeout 1500 -- "bash tools/prof_all.sh $batch 2>&1 | tail -4" 2>&1 | grep -E "gpurun\]|kernel_ms" | tail -4
  python tools/collect_profiles.py r06 $batch 2>&1 | tail -4
done
rm -rf gpurun_out/prof_*
ls profiles/r06
