#!/bin/bash
set -u
cd /root/repo
for batch in "config2 config2_lvrb config5_share"; do
  rm -rf gpurun_out/prof_*
  /usr/local/graft/bin/gpurun --timeout 1500 -- "bash tools/prof_all.sh $batch 2>&1 | tail -4" 2>&1 | grep -E "status=" | tail -2
  python tools/collect_profiles.py r06 $batch 2>&1 | tail -4
done
rm -rf gpurun_out/prof_*
