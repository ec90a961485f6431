#!/bin/bash
# FETCH_SIZE of tools/gather_probe's four access patterns against their known byte counts: tools/pmc_gather.sh <tag>
#   -> gpurun_out/pmc_gather_<tag>/ and a table on stdout (profiles/<tag>_gather_calibration.txt).  Counter pass and timing pass are separate.
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_gather_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
$REPO/tools/gather_probe 24 3 > $OUT/timing.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- $REPO/tools/gather_probe 24 2 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/rdreq -o pmc -- $REPO/tools/gather_probe 24 2 > $OUT/rdreq.log 2>&1
cat $OUT/timing.txt
python3 - <<PY
import sqlite3, glob, re
known = {"stream16": (2**24 * 128.0, 2**24 * 128.0), "gather_kernel<0, 5>": (2**24 * 80.0, 2**24 * 128.0), "gather_kernel<0, 8>": (2**24 * 128.0, 2**24 * 128.0), "gather_kernel<0, 4>": (2**24 * 64.0, 2**24 * 128.0)}
print("# FETCH_SIZE (KiB as reported, per launch) against the bytes the launch requested / the whole 128-byte lines it touched")
for sub in ("fetch", "rdreq"):
    for p in glob.glob("$OUT/%s/*.db" % sub):
        db = sqlite3.connect(p)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        cc = [t for t in tabs if t.startswith("counters_collection")][0]
        for name, counter, cnt, avg in db.execute(f"select kernel_name, counter_name, count(*), avg(value) from {cc} group by kernel_name, counter_name"):
            key = next((k for k in known if k.split("<")[0] in name and (("<" not in k) or k.split("<")[1].rstrip(">").replace(" ", "") in name.replace(" ", ""))), None)
            if key is None or "fill" in name:
                continue
            req, lines = known[key]
            if counter == "FETCH_SIZE":
                b = avg * 1024
                print(f"{key:22s} {counter:24s} launches {cnt:2d}  {avg:14.0f} KiB = {b:.4e} B   requested / reported = {req / b:.3f}   whole lines / reported = {lines / b:.3f}")
            else:
                print(f"{key:22s} {counter:24s} launches {cnt:2d}  {avg:14.0f} requests   requested bytes per request = {req / avg:.1f}   line bytes per request = {lines / avg:.1f}")
PY
