"""durations of the unit-path kernels per commitment batch from a rocprofv3 kernel-trace database: tools/unit_trace.py <trace dir>"""
import glob, re, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
rows = db.execute("select name, duration from kernels order by start").fetchall()
out = {}
for name, dur in rows:
    m = re.search(r"(msm_unit_\w+|msm_sparse_emit\w+)", name)
    if m:
        out.setdefault(m.group(1), []).append(round(dur / 1e3, 1))
for k, v in out.items():
    print(k, v)
