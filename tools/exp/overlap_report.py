"""Report of tools/exp/overlap_ab.sh: per-kernel durations, concurrency between the accumulation and the sort / reduction kernels, VALU instruction counts."""
import glob, re, sqlite3, sys
out = sys.argv[1]


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


SORT = ("msm_partition", "msm_bin_sort", "msm_bucket", "msm_seg_bucket", "msm_offsets", "msm_recode", "msm_zero")
TAIL = ("msm_combine", "msm_reduce", "msm_window_sum")


def klass(n):
    if n.startswith("msm_accumulate"):
        return "acc"
    if n.startswith(SORT):
        return "sort"
    if n.startswith(TAIL):
        return "tail"
    return "other"


for tag in ("v0", "g4_lean_acc2", "g4_regular_acc2", "g4_lean_acc3", "g4_regular_acc3"):
    dbs = glob.glob(f"{out}/{tag}/**/*.db", recursive=True)
    if not dbs:
        print(tag, "no trace"); continue
    db = sqlite3.connect(dbs[0])
    rows = [(short(r[0]), r[1], r[2]) for r in db.execute("select name, start, end from kernels order by start")]
    rows = [r for r in rows if r[0].startswith("msm_")]
    # the last five MSMs of the probe (the untimed wall-clock loop): split at msm_zero_ranges_kernel launches
    starts = [i for i, r in enumerate(rows) if r[0] == "msm_zero_ranges_kernel"]
    if len(starts) < 6:
        print(tag, "too few MSMs", len(starts)); continue
    spans = []
    per = {}
    conc = {"sort": 0.0, "tail": 0.0}
    for a, b in zip(starts[-6:-1], starts[-5:]):
        ks = rows[a:b]
        t0, t1 = min(k[1] for k in ks), max(k[2] for k in ks)
        spans.append((t1 - t0) / 1e6)
        acc_iv = [(k[1], k[2]) for k in ks if klass(k[0]) == "acc"]
        for k in ks:
            per.setdefault(k[0], [0, 0.0])
            per[k[0]][0] += 1; per[k[0]][1] += (k[2] - k[1]) / 1e6
            c = klass(k[0])
            if c in conc:
                for (s, e) in acc_iv:
                    lo, hi = max(s, k[1]), min(e, k[2])
                    if hi > lo:
                        conc[c] += (hi - lo) / 1e6
    n = len(spans)
    print(f"== {tag}: first-kernel-start .. last-kernel-end per MSM {sum(spans) / n:.3f} ms (5 MSMs: {' '.join(f'{x:.2f}' for x in spans)})")
    tot = {}
    for k, (cnt, ms) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:36s} launches/MSM {cnt / n:5.1f}  sum of durations {ms / n:8.3f} ms")
        tot[klass(k)] = tot.get(klass(k), 0) + ms / n
    print("   class sums (ms per MSM):", {k: round(v, 3) for k, v in tot.items()},
          " time a sort / tail kernel was in flight WHILE an accumulation launch was:", {k: round(v / n, 3) for k, v in conc.items()})
for tag in ("pmc_v0", "pmc_g4_lean"):
    dbs = glob.glob(f"{out}/{tag}/**/*.db", recursive=True)
    if not dbs:
        print(tag, "no counters"); continue
    db = sqlite3.connect(dbs[0])
    print(f"== {tag}: counters per launch (dispatches are serialised under --pmc), sums over the kernels of one MSM")
    agg = {}
    for name, ctr, cnt, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        k = short(name)
        if not k.startswith("msm_"):
            continue
        agg.setdefault(k, {})[ctr] = (cnt, avg)
    nm = None
    for k, d in agg.items():
        if k == "msm_zero_ranges_kernel":
            nm = list(d.values())[0][0]
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", (0, 0))[0] * kv[1].get("SQ_INSTS_VALU", (0, 0))[1]):
        per_msm = {c: v[0] * v[1] / nm for c, v in d.items()} if nm else {}
        print(f"   {k:36s} launches/MSM {list(d.values())[0][0] / nm:5.1f}  " + "  ".join(f"{c}={per_msm[c]:.4g}" for c in sorted(per_msm)))
