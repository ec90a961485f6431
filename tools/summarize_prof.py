#!/usr/bin/env python3
"""Turns the rocprofv3 rocpd databases written by tools/profile.sh into the committed summaries:
    profiles/<tag>_kernel_stats.md   (rocprofv3 --kernel-trace --stats)
    profiles/<tag>_pmc.md            (FETCH_SIZE / WRITE_SIZE passes, per kernel, per launch)
    profiles/traffic.json            (HBM bytes per launch that bench.py reports as roofline.traffic)
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and,
per MI355X_MICROARCH.md (HBM section), gfx950's FETCH_SIZE counts 128-B read requests at 64 B, so
the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated).
usage: tools/summarize_prof.py gpurun_out/prof_<tag> <tag> [msm_log_n ntt_log_n]
"""
import json
import os
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(\w+_kernel)\b", name)
    if m:
        t = re.search(r"<trh::(\w+)Params>", name)
        return m.group(1) + (f"<{t.group(1)}>" if t else "")
    return name.split("(")[0][-60:]


def provenance(src_dir, root, kernel_key):
    """{"build", "kernel", "vgpr", "isa_instructions"}: the library build the counters were collected on (version.txt written on the GPU box
    by tools/profile.sh / tools/pmc_sweep.sh) and the kernel's ISA as tools/isa_regs.py reads it from the same sources' objects"""
    import re as _re
    build = None
    try:
        m = _re.search(r"build ([0-9a-f]+)", open(os.path.join(src_dir, "version.txt")).read())
        build = m.group(1) if m else None
    except Exception:
        pass
    try:
        isa = json.load(open(os.path.join(root, "profiles", "isa_registers.json"))).get(kernel_key) or {}
    except Exception:
        isa = {}
    return {"build": build, "kernel": kernel_key, "vgpr": isa.get("vgpr"), "isa_instructions": isa.get("isa_instructions")}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    msm_log_n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    ntt_log_n = int(sys.argv[4]) if len(sys.argv) > 4 else 22
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    db = sqlite3.connect(os.path.join(src, "stats", "trace_results.db"))
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(grid_y), max(workgroup_x) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "",
             "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-check --no-sweep`",
             "(2 warm-up + 8 timed 2^%d Pallas MSMs, then the 2^%d Fp NTT loop).  Durations in microseconds." % (msm_log_n, ntt_log_n), "",
             "`VGPR (ISA)` is the kernel's .vgpr_count from the code object (tools/isa_regs.py -> profiles/isa_registers.json); `VGPR (rocprofv3)` is the",
             "trace's own column, an allocation-granule count on gfx950 (half the ISA figure for the kernels here).", "",
             "| kernel | calls | total us | avg us | min us | max us | % | VGPR (ISA) | VGPR (rocprofv3) | SGPR | LDS B | scratch | grid | wg |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    try:
        isa = json.load(open(os.path.join(out_dir, "isa_registers.json")))
    except Exception:
        isa = {}
    for r in rows:
        lines.append(f"| {short(r[0])} | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | {100*r[2]/total:.1f} | "
                     f"{(isa.get(short(r[0])) or isa.get(short(r[0]) + '<Fp>') or {}).get('vgpr', '')} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]}x{r[11]} | {r[12]} |")
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))

    wpath = os.path.join(src, "stats_witness", "trace_results.db")
    if os.path.exists(wpath):  # the witness-shaped replay: which kernels the skewed commitments really spend their time in
        wdb = sqlite3.connect(wpath)
        wrows = wdb.execute("select name, count(*), sum(duration), avg(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
        wtotal = sum(r[2] for r in wrows) or 1
        wl = [f"# rocprofv3 --kernel-trace --stats: k = 18 create_proof replay over WITNESS-SHAPED columns ({tag})", "",
              "Command: `rocprofv3 --kernel-trace --stats -- python3 tools/replay_probe.py 32 witness` (= python -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --no-keygen)",
              "(flags / 32-bit words on the n / 4 live rows, zero padding, blinding rows: a flag column is a plain sum of table entries (msm_unit_sum_kernel); the other classes fall into few buckets, where the",
              "chunked bucket passes (msm_bucket_pass_kernel) and the heavy-bucket combine (msm_combine_heavy_kernel) carry the commitments).  Durations in microseconds.", "",
              "| kernel | calls | total us | avg us | max us | % |", "|---|---|---|---|---|---|"]
        for r in wrows[:40]:
            wl.append(f"| {short(r[0])} | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {100*r[2]/wtotal:.1f} |")
        with open(os.path.join(out_dir, f"{tag}_witness_replay_kernel_stats.md"), "w") as fh:
            fh.write("\n".join(wl) + "\n")

    traffic_path = os.path.join(out_dir, "traffic.json")
    traffic = {}
    pmc = {}
    for kind, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        p = os.path.join(src, kind, "pmc_results.db")
        if not os.path.exists(p):
            continue
        d = sqlite3.connect(p)
        for name, grid, cnt, avg in d.execute("select kernel_name, grid_size, count(*), avg(value) from counters_collection "
                                               "where counter_name=? group by kernel_name, grid_size", (counter,)):
            pmc.setdefault((short(name), grid), {})[counter] = (cnt, avg)
    if pmc:
        lines = [f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) ({tag})", "",
                 "Per launch averages, KiB as reported; `HBM bytes` = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction).", "",
                 "| kernel | grid threads | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes / launch |", "|---|---|---|---|---|---|"]
        for (name, grid), v in sorted(pmc.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[1] + kv[1].get("WRITE_SIZE", (0, 0))[1])):
            f = v.get("FETCH_SIZE", (0, 0.0)); w = v.get("WRITE_SIZE", (0, 0.0))
            hbm = (2 * f[1] + w[1]) * 1024
            lines.append(f"| {name} | {grid} | {max(f[0], w[0])} | {f[1]:.0f} | {w[1]:.0f} | {hbm:.3e} |")
            if name.startswith("msm_accumulate") and hbm > (traffic.get(f"msm_accumulate_2^{msm_log_n}") or {}).get("bytes", 0):
                traffic[f"msm_accumulate_2^{msm_log_n}"] = dict(bytes=hbm, **provenance(src, root, "msm_accumulate_seg_kernel<Fp>"))  # the full-size launches (largest group)
        # the NTT is several launches of one kernel per transform: sum over the passes of one transform
        passes = 1 if ntt_log_n <= 11 else max(2, -(-ntt_log_n // 9))  # mirrors the pass plan in csrc/ntt.hip
        ntt_total = passes * max([(2 * v.get("FETCH_SIZE", (0, 0.0))[1] + v.get("WRITE_SIZE", (0, 0.0))[1]) * 1024
                                  for (name, grid), v in pmc.items() if name.startswith("ntt_pass")] or [0])
        if ntt_total:
            traffic[f"ntt_fp_2^{ntt_log_n}"] = dict(bytes=ntt_total, **provenance(src, root, "ntt_passy_kernel<Fp>"))
            lines += ["", f"NTT 2^{ntt_log_n}: HBM bytes per transform ({passes} passes x per-launch average) = {ntt_total:.3e} "
                          f"(algorithmic {64 * (1 << ntt_log_n):.3e})"]
        with open(os.path.join(out_dir, f"{tag}_pmc.md"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
        print("\n".join(lines))
        old = {}
        if os.path.exists(traffic_path):
            old = json.load(open(traffic_path))
        old.update(traffic)
        old["_source"] = f"profiles/{tag}_pmc.md"
        json.dump(old, open(traffic_path, "w"), indent=1)


if __name__ == "__main__":
    main()
