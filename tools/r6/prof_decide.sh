#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6/prof_decide_$1
mkdir -p $OUT
cd /tmp
python $R/tools/r6/decide_loop.py 50
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/tools/r6/decide_loop.py 30 > $OUT/trace.log 2>&1
timeout 60 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python $R/tools/r6/decide_loop.py 3 > $OUT/pmc1.log 2>&1
timeout 60 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/pmc2 -o p -- python $R/tools/r6/decide_loop.py 3 > $OUT/pmc2.log 2>&1
rm -f $OUT/trace/*/t_kernel_trace.csv $OUT/trace/t_kernel_trace.csv
python $R/tools/pmc_summary.py $(find $OUT/pmc1 $OUT/pmc2 -name "*counter_collection.csv") 2>/dev/null | grep -A9 "k_tlp_fast2" | head -40
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-220
