#!/usr/bin/env python3
"""profiles/<tag>_pmc_sweep.md and the sweep keys of profiles/traffic.json from tools/pmc_sweep.sh's counter passes.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The factor 2 is MI355X_MICROARCH.md's gfx950 correction for streaming reads AND,
since round 5, measured for this library's gather pattern as well (profiles/r05_gather_calibration.txt: every fabric read request is a whole
128-byte line tallied at 64 bytes, TCC_EA0_RDREQ_32B = 0, for 16-byte-per-lane streams and for 80-byte gathers out of random 128-byte records
alike).  WRITE_SIZE is taken as reported.
usage: tools/summarize_sweep_pmc.py gpurun_out/pmc_sweep_<tag> <tag>"""
import glob, json, os, re, sqlite3, sys


def per_kernel(path, counter):
    out = {}
    for p in glob.glob(os.path.join(path, "*.db")):
        db = sqlite3.connect(p)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        cc = [t for t in tabs if t.startswith("counters_collection")][0]
        for name, grid, cnt, avg in db.execute(f"select kernel_name, grid_size, count(*), avg(value) from {cc} where counter_name=? group by kernel_name, grid_size", (counter,)):
            m = re.search(r"(\w+_kernel)", name)
            out[(m.group(1) if m else name[:40], grid)] = (cnt, avg)
    return out


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
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = [f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE at the sizes of bench.py's sweep ({tag}; tools/pmc_sweep.sh, one counter per pass)", "",
             "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (factor 2: every fabric read request is a 128-byte line tallied at 64 bytes -- the guide's streaming",
             "calibration, confirmed for this library's gathers in profiles/r05_gather_calibration.txt).", "",
             "| workload | kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes / launch | algorithmic bytes / launch | ratio |", "|---|---|---|---|---|---|---|---|"]
    traffic = {}
    for lg in (20, 22, 26):
        f = per_kernel(os.path.join(src, f"msm{lg}_fetch"), "FETCH_SIZE")
        w = per_kernel(os.path.join(src, f"msm{lg}_write"), "WRITE_SIZE")
        acc = {k: v for k, v in f.items() if k[0].startswith("msm_accumulate")}
        if not acc:
            continue
        key = max(acc, key=lambda k: acc[k][1])  # the full-size launches
        hbm = (2 * acc[key][1] + w.get(key, (0, 0.0))[1]) * 1024
        per_launch = (1 << lg) if lg <= 25 else (1 << 25)
        traffic[f"msm_accumulate_2^{lg}"] = dict(bytes=hbm, **provenance(src, root, "msm_accumulate_seg_kernel<Fp>"))
        lines.append(f"| Pallas MSM 2^{lg} | {key[0]} | {acc[key][0]} | {acc[key][1]:.0f} | {w.get(key, (0, 0.0))[1]:.0f} | {hbm:.3e} | {96.0 * per_launch:.3e} | {hbm / (96.0 * per_launch):.1f} |")
    for lg in (20, 24):
        f = per_kernel(os.path.join(src, f"ntt{lg}_fetch"), "FETCH_SIZE")
        w = per_kernel(os.path.join(src, f"ntt{lg}_write"), "WRITE_SIZE")
        passes = {k: v for k, v in f.items() if k[0].startswith("ntt_pass")}
        if not passes:
            continue
        # launches of one transform: every pass kernel launch / number of transforms run (the probe runs 2 warm-up + 5 timed + 1 for the output digest = 8 transforms)
        n_launch = sum(v[0] for v in passes.values())
        total = sum(v[0] * (2 * v[1] + w.get(k, (0, 0.0))[1]) * 1024 for k, v in passes.items())
        transforms = 8
        hbm = total / transforms
        traffic[f"ntt_fp_2^{lg}"] = dict(bytes=hbm, **provenance(src, root, "ntt_passy_kernel<Fp>"))
        lines.append(f"| Fp NTT 2^{lg} | ntt_passy_kernel x {n_launch // transforms} passes | {n_launch} | {sum(v[0] * v[1] for v in passes.values()) / transforms:.0f} | "
                     f"{sum(w.get(k, (0, 0.0))[0] * w.get(k, (0, 0.0))[1] for k in passes) / transforms:.0f} | {hbm:.3e} | {64.0 * (1 << lg):.3e} | {hbm / (64.0 * (1 << lg)):.1f} |")
    with open(os.path.join(root, "profiles", f"{tag}_pmc_sweep.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    tp = os.path.join(root, "profiles", "traffic.json")
    old = json.load(open(tp)) if os.path.exists(tp) else {}
    old.update(traffic)
    old["_source_sweep"] = f"profiles/{tag}_pmc_sweep.md"
    json.dump(old, open(tp, "w"), indent=1)


if __name__ == "__main__":
    main()
