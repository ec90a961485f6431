#!/bin/bash
# SQ counters of the two forms of the signed NTT pass, same box: tools/pmc_ntt_pipe.sh <tag> [log_n] [batch]
#   -> gpurun_out/pmc_ntt_pipe_<tag>/ and a table on stdout.  Two counter passes per form (8 SQ slots), kernel trace only beside them.
set -u
TAG=${1:-r05}
LOGN=${2:-22}
BATCH=${3:-1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_ntt_pipe_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for PIPE in 0 1; do
  export TRH_NTT_PIPE=$PIPE
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/p${PIPE}a -o pmc -- python3 $REPO/tools/ntt_probe.py $LOGN 5 $BATCH > $OUT/p${PIPE}a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -d $OUT/p${PIPE}b -o pmc -- python3 $REPO/tools/ntt_probe.py $LOGN 5 $BATCH > $OUT/p${PIPE}b.log 2>&1
done
python3 - <<PY
import sqlite3, glob
print("# signed NTT pass, 2^$LOGN x $BATCH Fp: ntt_passy_kernel (TRH_NTT_PIPE=0) vs the persistent pipelined ntt_passp_kernel (TRH_NTT_PIPE=1); sums over a transform's passes, averaged over the probe's transforms (tools/pmc_ntt_pipe.sh $TAG $LOGN $BATCH)")
for pipe in (0, 1):
    for part in "ab":
        for p in sorted(glob.glob("$OUT/p%d%s/*.db" % (pipe, part))):
            db = sqlite3.connect(p)
            tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
            cc = [t for t in tabs if t.startswith("counters_collection")][0]
            rows = list(db.execute(f"select counter_name, count(*), sum(value) from {cc} where kernel_name like '%ntt_pass%' group by counter_name"))
            for name, cnt, tot in rows:
                print(f"pipe={pipe} {name:28s} launches {cnt:4d}  per-launch {tot / cnt:.6g}")
            if part == "a":
                kk = [t for t in tabs if t == "kernels"]
                if kk:
                    for row in db.execute("select count(*), avg(duration) from kernels where name like '%ntt_pass%'"):
                        print(f"pipe={pipe} {'duration_ns (counter run)':28s} launches {row[0]:4d}  per-launch {row[1]:.6g}")
PY
tail -2 $OUT/p0a.log $OUT/p1a.log
