#!/bin/bash
# PMC passes of one workload's sweep kernels with extra bench options: tools/r6/pmc_nrt.sh <workload> <tag> [bench options...]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; TAG=$2; shift 2
OUT=$R/gpurun_out/r6/pmc_${W}_$TAG
mkdir -p $OUT
cd /tmp
B="python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 3 --warmup 1 $*"
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --workload $W --cpu-budget 0 --sweep-only --no-every-row --steps 30 --warmup 5 $* > $OUT/trace.log 2>&1
timeout 40 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- $B > $OUT/pmc1.log 2>&1
timeout 40 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/pmc2 -o p -- $B > $OUT/pmc2.log 2>&1
timeout 40 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc3 -o p -- $B > $OUT/pmc3.log 2>&1
rm -f $OUT/trace/*/t_kernel_trace.csv $OUT/trace/t_kernel_trace.csv
python $R/tools/pmc_summary.py $(find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 -name "*counter_collection.csv") 2>/dev/null | grep -A12 "k_nrt\|k_rows" | head -120
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs head -8
