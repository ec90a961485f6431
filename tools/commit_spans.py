"""Per-batch wall spans of the commitment batches of the k = 18 witness replay, from a rocprofv3 kernel trace database:
tools/commit_spans.py <trace dir>   (the batches are found by their first kernel: msm_sparse_sample_kernel or msm_recode_kernel with grid z >= 6)"""
import sqlite3, glob, re, sys
p = glob.glob(sys.argv[1] + "/*.db")[0]
db = sqlite3.connect(p)
rows = db.execute("select name, start, end, duration, grid_x, grid_z from kernels order by start").fetchall()
def short(n):
    m = re.search(r"(msm_\w+|ntt_\w+|\w+_kernel|__amd\w+)", n)
    return m.group(1) if m else n[:30]
names = [short(r[0]) for r in rows]
starts = []
i = 0
while i < len(rows):
    prev = i - 1
    while prev >= 0 and names[prev].startswith("__amd"):
        prev -= 1
    if names[i].startswith("msm_sparse_sample") or (names[i].startswith("msm_recode") and (prev < 0 or not names[prev].startswith("msm_"))):
        j = i; busy = 0; per = {}
        while j < len(rows) and not names[j].startswith(("ntt_", "powers", "lincomb", "ipa_")):
            busy += rows[j][3]; per[names[j]] = per.get(names[j], 0) + rows[j][3]; j += 1
        span = (rows[j - 1][2] - rows[i][1]) / 1e3
        z = max(r[5] for r in rows[i:j])
        top = sorted(per.items(), key=lambda kv: -kv[1])[:6]
        print(f"z={z:3d} span {span:9.1f} us busy {busy/1e3:9.1f} idle {span-busy/1e3:8.1f} | " + ", ".join(f"{k.replace('msm_','').replace('_kernel','')} {v/1e3:.0f}" for k, v in top))
        starts.append(span)
        i = j
    else:
        i += 1
print("sum of spans %.1f us" % sum(starts))
