#!/bin/bash
# SQ counters of the NTT pass kernel: tools/pmc_ntt.sh <tag>  -> gpurun_out/pmc_ntt_<tag>/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_ntt_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export TRH_SELFTEST=0  # the self-test's own small launches (2^10 MSMs, 2^10 / 2^12 transforms) would be averaged into the per-kernel figures
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/sq -o pmc -- python3 $REPO/tools/ntt_probe.py 22 5 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES -d $OUT/sq2 -o pmc -- python3 $REPO/tools/ntt_probe.py 22 5 > $OUT/sq2.log 2>&1
python3 - <<PY
import sqlite3, glob
for d in ("sq", "sq2"):
    for p in glob.glob("$OUT/%s/*.db" % d):
        db = sqlite3.connect(p)
        for row in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%ntt_pass%' group by kernel_name, counter_name"):
            print(row[0][:60], row[1], row[2], row[3])
PY
