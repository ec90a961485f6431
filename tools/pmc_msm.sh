#!/bin/bash
# SQ counters of the MSM accumulate kernel: tools/pmc_msm.sh <tag> [log_n]  -> gpurun_out/pmc_msm_<tag>/
set -u
TAG=${1:-r01}
LOGN=${2:-24}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_msm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export TRH_SELFTEST=0  # the self-test's own small launches (2^10 MSMs, 2^10 / 2^12 transforms) would be averaged into the per-kernel figures
cd /tmp
COUNTERS=${TRH_PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES}
rocprofv3 --kernel-trace --pmc $COUNTERS -d $OUT/sq -o pmc -- python3 $REPO/tools/msm_probe.py $LOGN pallas 0 0 > $OUT/sq.log 2>&1
python3 - <<PY
import sqlite3, glob
for p in glob.glob("$OUT/sq/*.db"):
    db = sqlite3.connect(p)
    for row in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%msm_accumulate%' group by kernel_name, counter_name"):
        print(row[0][40:90], row[1], row[2], row[3])
PY
