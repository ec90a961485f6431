#!/bin/bash
# rocprofv3 kernel trace of an arbitrary probe: tools/prof_cmd.sh <tag> <script.py> [args...]  -> per-kernel table on stdout
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export TRH_SELFTEST=0  # the self-test's own small launches (2^10 MSMs, 2^10 / 2^12 transforms) would be averaged into the per-kernel figures
SCRIPT=$REPO/$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python3 $SCRIPT "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import sqlite3, glob, re
for p in glob.glob("$OUT/*.db"):
    db = sqlite3.connect(p)
    rows = db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    for r in rows[:16]:
        m = re.search(r"(\w+_kernel)", r[0])
        print(f"{(m.group(1) if m else r[0][:40]):36s} calls {r[1]:5d}  total {r[2]/1e6:9.3f} ms  avg {r[3]/1e3:9.1f} us  {100*r[2]/tot:5.1f}%")
PY
