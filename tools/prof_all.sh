# every rocprofv3 invocation runs under `timeout`: one hung pass (seen once, config5_share under --kernel-trace) must not eat the GPU budget
# (--no-every-row: bench.py's every_row section dispatches the same kernels with pod classes off: it would skew the per-dispatch means)
# collects rocprofv3 kernel-trace stats + PMC (separate passes) for the bench workloads -> gpurun_out/prof_<workload>/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for W in "$@"; do
  OUT=$R/gpurun_out/prof_$W
  mkdir -p $OUT
  cd /tmp
  # the kernel-trace pass runs the default bench command (100 steps / 20 warmup) so that rocprofv3's average launch
  # duration and bench.py's own HIP-event figure describe the same sustained-clock regime; long sweeps use fewer steps
  # config5_share's full_cycle section replays the single-row kernels 62.5k times (sequential commit): kept out of the stats
  # config2 (the default line): without the other workloads' legs and the config #5 leg — their kernels are not this workload's
  case $W in config3_leastnuma|config5) ST="--steps 10 --warmup 2";; config5_share) ST="--sweep-only";; config2) ST="--no-legs --no-config5-leg";; *) ST="";; esac
  timeout 45 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --workload $W --cpu-budget 0 --no-every-row $ST > $OUT/trace.log 2>&1
  timeout 30 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 3 --warmup 1 > $OUT/pmc1.log 2>&1
  timeout 30 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/pmc2 -o p -- python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 3 --warmup 1 > $OUT/pmc2.log 2>&1
  timeout 30 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -o p -- python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 3 --warmup 1 > $OUT/pmc3.log 2>&1
  timeout 30 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc4 -o p -- python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 3 --warmup 1 > $OUT/pmc4.log 2>&1
  rm -f $OUT/trace/t_kernel_trace.csv  # the per-launch list: megabytes per workload, and gpurun brings back at most 64 MiB; the stats stay
  tail -1 $OUT/trace.log | cut -c1-200
done
