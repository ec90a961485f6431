#!/bin/bash
# VALU-busy question of msm_accumulate_seg_kernel: tools/pmc_valu.sh <tag> [log_n] -> gpurun_out/pmc_valu_<tag>/
# Two counter passes (SQ has 8 slots), kernel trace only beside them.
set -u
TAG=${1:-r03}
LOGN=${2:-24}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_valu_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export TRH_SELFTEST=0  # the self-test's own small launches (2^10 MSMs, 2^10 / 2^12 transforms) would be averaged into the per-kernel figures
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $OUT/a -o pmc -- python3 $REPO/tools/msm_probe.py $LOGN pallas 0 0 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_CYCLES SQ_WAVES -d $OUT/b -o pmc -- python3 $REPO/tools/msm_probe.py $LOGN pallas 0 0 > $OUT/b.log 2>&1
python3 - <<PY
import sqlite3, glob
for p in sorted(glob.glob("$OUT/*/*.db")):
    db = sqlite3.connect(p)
    for row in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%msm_accumulate_seg%' group by kernel_name, counter_name"):
        print(row[0][40:90], row[1], row[2], row[3])
    try:
        for row in db.execute("select name, count(*), avg(duration) from kernels where name like '%msm_accumulate_seg%' group by name"):
            print("duration_ns", row[0][40:80], row[1], row[2])
    except Exception as e:
        print("no kernels view", e)
PY
