#!/bin/bash
# Kernel timeline of the k = 18 opening around the generator fold: tools/exp/fold_trace.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp TRH_SELFTEST=0
cd /tmp
rm -rf $REPO/gpurun_out/fold_tr
rocprofv3 --kernel-trace -d $REPO/gpurun_out/fold_tr -o t -- python3 $REPO/tools/ipa_probe.py 18 > /dev/null 2>&1
python3 - <<PY
import glob, re, sqlite3
for p in glob.glob("$REPO/gpurun_out/fold_tr/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "ipa_fold_accumulate" in r[0]]
    if not idx: print("no fold"); continue
    a = idx[-1]
    # from two front kernels before the fold to three after
    fr = [i for i, r in enumerate(rows) if "ipa_round_front" in r[0]]
    start = max(i for i in fr if i < a)
    start = max([i for i in fr if i < start] or [start])
    after = [i for i in fr if i > a][:5]
    end = after[-1] if after else len(rows) - 1
    t0 = rows[start][1]; prev = None
    for name, s, e, sid in rows[start:end + 1]:
        m = re.search(r"(\w+_kernel)", name)
        gap = (s - prev) / 1e3 if prev else 0
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  st {sid}  {m.group(1) if m else name[:40]}")
        prev = e
PY
python3 $REPO/tools/exp/opening_phases.py $REPO/gpurun_out/fold_tr 18
