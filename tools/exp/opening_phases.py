"""Where the time of the last opening in a rocprofv3 kernel trace went: tools/exp/opening_phases.py <dir with *.db> <k>"""
import glob, re, sqlite3, sys
d, k = sys.argv[1], int(sys.argv[2])
p = glob.glob(d + "/**/*.db", recursive=True)[0]
rows = sqlite3.connect(p).execute("select name, start, end from kernels order by start").fetchall()
fr = [i for i, r in enumerate(rows) if "ipa_round_front" in r[0]][-k:]
ev = [i for i, r in enumerate(rows) if "ipa_combine_eval" in r[0]]
pro = max(i for i in ev if i < fr[0]) - 0
pro = max(i for i in ev if i < pro)  # the first of the two
while pro > 0 and "powers" not in rows[pro][0]: pro -= 1
end = fr[-1]
while end + 1 < len(rows) and "ipa_round_update" not in rows[end][0]: end += 1
t = lambda i: rows[i][1] / 1e3
print(f"prologue (powers .. first front launch): {t(fr[0]) - t(pro):8.1f} us")
fold = [i for i, r in enumerate(rows) if "ipa_fold_accumulate" in r[0] and fr[0] < i < end]
for j in range(k):
    a, b = fr[j], (fr[j + 1] if j + 1 < k else end)
    names = " ".join(sorted({re.search(r"(\w+)_kernel", r[0]).group(1) for r in rows[a:b] if re.search(r"(\w+)_kernel", r[0]) and ("fold" in r[0] or "small" in r[0])}))
    busy = sum(r[2] - r[1] for r in rows[a:b]) / 1e3
    print(f"round {j:2d}: {t(b) - t(a):8.1f} us  (kernels {busy:7.1f})  {names}")
print(f"rounds total {t(end) - t(fr[0]):8.1f} us; opening {rows[end][2] / 1e3 - t(pro):8.1f} us")
