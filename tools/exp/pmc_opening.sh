#!/bin/bash
# SQ counters of the kernels of a k = 18 opening (the collapse's three kernels, msm_small_kernel, the full-size round's tail):
#   tools/exp/pmc_opening.sh  -> per-kernel averages on stdout.  Counters only (no trace domains beside the kernel trace).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_opening
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp TRH_SELFTEST=0
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/sq -o pmc -- python3 $REPO/tools/ipa_probe.py 18 > $OUT/sq.log 2>&1
python3 - <<PY
import sqlite3, glob, re
for p in glob.glob("$OUT/sq/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    by = {}
    for k, c, n, v in rows:
        m = re.search(r"(\w+_kernel)", k)
        by.setdefault(m.group(1) if m else k[:40], {})[c] = (n, v)
    want = ["ipa_fold_accumulate_kernel", "ipa_fold_reduce_kernel", "ipa_fold_affine_kernel", "msm_small_kernel", "msm_accumulate_seg_kernel", "msm_reduce_q4_kernel", "msm_combine_q4_kernel", "msm_window_sum_q4_kernel", "ipa_round_front_kernel"]
    print("kernel | launches | waves | VALU wave-instructions | GUI active cycles | SQ busy cycles | wave cycles | wait-inst cycles | VALU issue utilisation = VALU wave-instructions x 4 cycles / (GUI active cycles / 8 XCDs x 1024 SIMDs)")
    for k in want:
        if k not in by: continue
        c = by[k]
        g = lambda n: c.get(n, (0, 0.0))[1]
        gui = g("GRBM_GUI_ACTIVE")
        print(f"{k} | {c.get('SQ_WAVES', (0, 0))[0]} | {g('SQ_WAVES'):.0f} | {g('SQ_INSTS_VALU'):.3e} | {gui:.3e} | {g('SQ_BUSY_CYCLES'):.3e} | {g('SQ_WAVE_CYCLES'):.3e} | {g('SQ_WAIT_INST_ANY'):.3e} | {g('SQ_INSTS_VALU') * 4 / (gui / 8 * 1024) if gui else 0:.2f}")
PY
