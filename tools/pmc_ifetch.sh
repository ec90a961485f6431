#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_ifetch
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU -d $OUT/ntt -o pmc -- python3 $REPO/tools/ntt_probe.py 22 5 > $OUT/ntt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU -d $OUT/msm -o pmc -- python3 $REPO/tools/msm_probe.py 22 pallas 0 0 > $OUT/msm.log 2>&1
python3 - <<PY
import sqlite3, glob
for d in ("ntt","msm"):
    for p in glob.glob("$OUT/"+d+"/*.db"):
        db = sqlite3.connect(p)
        for row in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%ntt_passy%' or kernel_name like '%msm_accumulate%' group by kernel_name, counter_name"):
            print(row[0][30:70], row[1], row[2], row[3])
PY
