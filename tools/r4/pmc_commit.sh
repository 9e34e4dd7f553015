#!/bin/bash
# PMC counters of the sequential commit loop's kernel (k_commit_trimaran_reg, one workgroup): where do a pod's 3 700 cycles go?
#   tools/r4/pmc_commit.sh        -> gpurun_out/pmc_commit/{pmc1,pmc2}/  + a summary on stdout
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/pmc_commit; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python $R/tools/r4/time_commit_trimaran.py 10000 100000 0,1 0 1 > $OUT/pmc1.log 2>&1
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc2 -o p -- python $R/tools/r4/time_commit_trimaran.py 10000 100000 0,1 0 1 > $OUT/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(dict)
for f in glob.glob("/root/repo/gpurun_out/pmc_commit/pmc*/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/pmc_commit/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_commit_trimaran_reg" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-24s %.4g (x%d)" % (n, sum(v) / len(v), len(v)))
PY
