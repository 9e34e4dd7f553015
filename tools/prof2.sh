# usage: bash tools/prof2.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ : issue/stall counters of the sweep kernels
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VALU --output-format csv -d $OUT/pmcA -o p -- python $R/bench.py --cpu-budget 0 --steps 3 --warmup 1 "$@" > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_LEVEL_WAVES SQ_WAIT_ANY --output-format csv -d $OUT/pmcB -o p -- python $R/bench.py --cpu-budget 0 --steps 3 --warmup 1 "$@" > $OUT/pmcB.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_CYCLES --output-format csv -d $OUT/pmcC -o p -- python $R/bench.py --cpu-budget 0 --steps 3 --warmup 1 "$@" > $OUT/pmcC.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmcA/p_counter_collection.csv $OUT/pmcB/p_counter_collection.csv $OUT/pmcC/p_counter_collection.csv
