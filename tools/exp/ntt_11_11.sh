#!/bin/bash
# EXPERIMENT (round 6): 2^22 as two passes of 11 stages against the shipping 8 + 7 + 7
# NOTE: drives the TRH_EXP_NTT_* knobs of the experiment build; the library no longer has them -- kept as the record of how profiles/r06_ntt_11_11_ab.txt was made
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" python3 tools/ntt_probe.py 22 20 2>&1 | tail -1; }
for pass in 1 2; do
run A=0
run TRH_EXP_NTT_PLAN=11,11
run TRH_EXP_NTT_PLAN=11,11 TRH_EXP_NTT_XCD=1
run TRH_EXP_NTT_PLAN=10,6,6
run TRH_EXP_NTT_PLAN=10,12
done
cd /tmp
pmc() { tag=$1; shift; for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  env "$@" rocprofv3 --kernel-trace --pmc $ctr -d $REPO/gpurun_out/ntt11_$tag -o p -- python3 $REPO/tools/ntt_probe.py 22 5 > /dev/null 2>&1
  python3 - <<PY
import sqlite3, glob
for p in glob.glob("$REPO/gpurun_out/ntt11_$tag/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    for row in db.execute("select counter_name, grid_size, count(*), avg(value) from counters_collection where kernel_name like '%ntt_passy%' group by counter_name, grid_size"):
        print("$tag", row)
PY
  rm -rf $REPO/gpurun_out/ntt11_$tag
done; }
pmc v0 A=0
pmc p1111 TRH_EXP_NTT_PLAN=11,11
pmc p1111x TRH_EXP_NTT_PLAN=11,11 TRH_EXP_NTT_XCD=1
