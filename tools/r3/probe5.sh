#!/bin/bash
export TMPDIR=/tmp SPX_NRT_CPB=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/pmc; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $O/avail.txt
for q in "0,0,1" "0,1,0" "1,0,0"; do
  n=${q//,/}
  timeout 60 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $O/a$n -o p -- python $R/tools/r3/exp_one.py LeastAllocated $q > $O/a$n.log 2>&1
  timeout 60 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/b$n -o p -- python $R/tools/r3/exp_one.py LeastAllocated $q > $O/b$n.log 2>&1
  timeout 60 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/c$n -o p -- python $R/tools/r3/exp_one.py LeastAllocated $q > $O/c$n.log 2>&1
  echo "== qos $q"; tail -1 $O/a$n.log
  python $R/tools/r3/pmc.py $O/a$n; python $R/tools/r3/pmc.py $O/b$n; python $R/tools/r3/pmc.py $O/c$n
done
cat $O/avail.txt | head -c 3000
find $O -name "*.csv" -size +200k -delete
