#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/pmc3; rm -rf $O; mkdir -p $O
cd /tmp
S=LeastNUMANodes; q="1,0,0"; n=100
timeout 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY --output-format csv -d $O/a$n -o p -- python $R/tools/r3/exp_one.py $S $q 2 > $O/a$n.log 2>&1
timeout 90 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b$n -o p -- python $R/tools/r3/exp_one.py $S $q 2 > $O/b$n.log 2>&1
tail -1 $O/a$n.log
python $R/tools/r3/pmc.py $O/a$n; python $R/tools/r3/pmc.py $O/b$n
find $O -name "*.csv" -size +200k -delete
