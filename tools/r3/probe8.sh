#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/pmc2; rm -rf $O; mkdir -p $O
cd /tmp
S=${1:-LeastAllocated}
for q in "0.5,0.4,0.1" "1,0,0"; do
  n=${q//[,.]/}
  timeout 60 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY --output-format csv -d $O/a$n -o p -- python $R/tools/r3/exp_one.py $S $q > $O/a$n.log 2>&1
  timeout 60 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/b$n -o p -- python $R/tools/r3/exp_one.py $S $q > $O/b$n.log 2>&1
  echo "== qos $q"; tail -1 $O/a$n.log
  python $R/tools/r3/pmc.py $O/a$n; python $R/tools/r3/pmc.py $O/b$n
done
find $O -name "*.csv" -size +200k -delete
