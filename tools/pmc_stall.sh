#!/bin/bash
# Where msm_accumulate_seg_kernel's non-issuing cycles go: tools/pmc_stall.sh <tag> [log_n]
#   -> gpurun_out/pmc_stall_<tag>/ and a table on stdout (copied to profiles/<tag>_msm_stall_counters.txt)
# Three counter passes (SQ has 8 slots; SQC shares them), kernel trace only beside them, never a trace domain.
set -u
TAG=${1:-r04}
LOGN=${2:-24}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_stall_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export TRH_SELFTEST=0  # the self-test's own small launches (2^10 MSMs, 2^10 / 2^12 transforms) would be averaged into the per-kernel figures
cd /tmp
P="python3 $REPO/tools/msm_probe.py $LOGN pallas 0 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $OUT/a -o pmc -- $P > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAVES -d $OUT/b -o pmc -- $P > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL -d $OUT/c -o pmc -- $P > $OUT/c.log 2>&1
python3 - <<PY
import sqlite3, glob
print("# msm_accumulate_seg_kernel, 2^$LOGN Pallas, averages per launch over the probe's launches (tools/pmc_stall.sh $TAG $LOGN)")
for p in sorted(glob.glob("$OUT/*/*.db")):
    db = sqlite3.connect(p)
    for row in db.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%msm_accumulate_seg%' group by counter_name"):
        print(f"{row[0]:32s} launches {row[1]:3d}  avg {row[2]:.6g}")
    try:
        for row in db.execute("select count(*), avg(duration) from kernels where name like '%msm_accumulate_seg%'"):
            print(f"{'duration_ns':32s} launches {row[0]:3d}  avg {row[1]:.6g}")
    except Exception as e:
        print("no kernels view", e)
PY
tail -3 $OUT/a.log $OUT/b.log $OUT/c.log
