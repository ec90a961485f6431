"""Kernel timeline between two occurrences of a marker kernel in a rocprofv3 kernel trace (rocpd sqlite): one IPA round, one lone MSM, ...
    tools/round_timeline.py <dir with *.db> <marker kernel substring> [which occurrence counted from the end, default 6]"""
import glob, re, sqlite3, sys
d, marker = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 6
for p in glob.glob(d + "/*.db"):
    db = sqlite3.connect(p)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(idx) < back + 1:
        print("marker seen", len(idx), "times only"); continue
    a, b = idx[-back - 1], idx[-back]
    t0 = rows[a][1]
    busy = 0
    prev_end = None
    print(f"# {b - a} kernels between occurrence -{back + 1} and -{back} of {marker}: {(rows[b][1] - t0) / 1e3:.1f} us")
    for name, s, e in rows[a:b]:
        m = re.search(r"(\w+_kernel)", name)
        short = m.group(1) if m else name[:40]
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  {short}")
        busy += e - s
        prev_end = e
    print(f"# busy {busy / 1e3:.1f} us, idle {(rows[b][1] - t0 - busy) / 1e3:.1f} us (last gap to the next marker: {(rows[b][1] - prev_end) / 1e3:.1f} us)")
