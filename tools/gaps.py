"""Busy / idle split of the GPU timeline in a rocprofv3 kernel trace (rocpd sqlite): tools/gaps.py <dir with *.db> [kernel name filter for the window]"""
import glob, sqlite3, sys
for p in glob.glob(sys.argv[1] + "/*.db"):
    db = sqlite3.connect(p)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        continue
    # restrict to the second half of the run (the warm repetition)
    rows = rows[len(rows) // 2:]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    busy = 0
    cur_s, cur_e = rows[0][1], rows[0][2]
    for _, s, e in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"{len(rows)} kernels over {(t1 - t0) / 1e6:.3f} ms: busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
