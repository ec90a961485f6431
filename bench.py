#!/usr/bin/env python3
"""Headline benchmark: Pallas MSM pairs/s @ 2^24 (+ Fp NTT elems/s @ 2^22) on 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path: a Pallas multi-scalar multiplication over 2^24 (scalar, base)
pairs per GPU, inputs already resident in HBM.  With N GPUs the global MSM has N * 2^24 pairs,
range-sharded across ranks (weak scaling); each rank runs its local Pippenger to one partial
point, the 96-byte partials are all-gathered over RCCL and summed on every rank (EC addition is
not an RCCL reduce op).  Rank 0 prints ONE JSON line.

`roofline` is for the dominant kernel (msm_accumulate_seg_kernel): algorithmic bytes = 96 B per pair
(32 B scalar + 64 B base) x pairs per launch, divided by that kernel's average duration measured
with HIP events on the launch stream inside the timed region (libtrh's timing hooks).
`cpu_baseline` times oracle/cpu_ref.cpp (the C++ restatement of halo2_proofs' rayon
best_multiexp; kind "port") on a bounded sample on this box's host cores.

Outside the timed headline region the same process also reports
  sweep            (N = 1) MSM at 2^20 / 2^22 / 2^26 and NTT at 2^20 / 2^24, each checked (closed form / inverse round trip) with
                   its own roofline fraction -- north_star's 2^20 .. 2^26 size range;
  strong           (N > 1) BASELINE config 5: ONE 2^26 Pallas MSM range-sharded over the N ranks (2^26 / N pairs each), same
                   RCCL all-gather of the 96-byte partials;
  single_process   (N > 1) the same 2^26 MSM through the C ABI's device group from rank 0 alone (trh_init_multi over the N GPUs:
                   what a single Rust prover process linking libtrh.so gets), device-resident scalars handed over with peer copies.
  e2e              (N = 1) BASELINE config 4 under the driver's clock: the k = 18 witness-shaped create_proof schedule replay
                   (tiny_ram_halo2_amd.replay) in three modes -- polynomials resident, host polynomials through the batched
                   host-pointer entries, host polynomials through one trh_msm / trh_best_fft call at a time (north_star's literal
                   integration) -- with GPU / PCIe-inclusive wall totals, link GB/s against the probed peak and the number of
                   primitive results compared with the oracle (never the thing timed).  --no-e2e skips it;
  collective       which backend carried the partials, over how many ranks and which devices, proven by an all-reduce.
`python bench.py --gpus N` WITHOUT a torch.distributed.run environment spawns its N ranks itself (fresh child processes, started
before this process makes any GPU call; the parent only waits and relays rank 0's line).
`--global-log-n L` makes the HEADLINE run strong-scaling instead (2^L pairs in total, split over the ranks); the JSON line
says which mode produced `value`.  A failed result check prints the line with "check": "MISMATCH" and exits with status 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Pallas MSM pairs/s @ 2^24 + Fp NTT elems/s @ 2^22; 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def usable_cores() -> int:
    """cores this process may actually run on (affinity mask and cgroup quota), not the box total"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline_msm(curve: str, log_cap: int = 24):
    """oracle leg: chunk-per-thread Pippenger restatement on the host cores, bounded sample (~10-25 s)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ref  # oracle -- the thing timed here, never the product path
    from tiny_ram_halo2_amd import synth

    threads = usable_cores()
    # calibrate on 2^18 pairs (2^16 under-estimates the rate of a 16-thread run by a third), then size one MSM for <= ~16 s -- the headline
    # 2^24 on a 16-core box, so that the oracle comparison below happens at the headline size -- and repeat it until >= 10 s have been spent
    n0 = 1 << 18
    bases0 = cpu_ref.gen_bases(curve, synth.BASE_S0, synth.BASE_D, n0, threads)
    sc0 = synth.msm_scalars(18)
    t = time.perf_counter()
    cpu_ref.best_multiexp(curve, sc0, bases0, threads)
    rate0 = n0 / (time.perf_counter() - t)
    log_n = 18
    while log_n < log_cap and (1 << (log_n + 1)) / rate0 < 16.0:
        log_n += 1
    n = 1 << log_n
    bases = cpu_ref.gen_bases(curve, synth.BASE_S0, synth.BASE_D, n, threads)
    sc = synth.msm_scalars(log_n)
    reps, t = 0, time.perf_counter()
    while True:
        last = cpu_ref.best_multiexp(curve, sc, bases, threads)
        reps += 1
        dt = time.perf_counter() - t
        if dt >= 10.0 or reps >= 8:
            break
    # the oracle's point for exactly the headline inputs (scalars synth.msm_scalars(log_n), bases (s0 + i d) G): the caller compares
    # the GPU result with it limb for limb, outside every timed region
    return ({"value": n * reps / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
             "sample": f"{reps} x 2^{log_n} Pallas best_multiexp (oracle/cpu_ref.cpp, {threads} threads), {dt:.2f} s"}, log_n, cpu_ref.to_affine(curve, last))


def cpu_baseline_ntt(field: str, log_n: int):
    import cpu_ref
    import pasta as o
    from tiny_ram_halo2_amd import synth

    threads = usable_cores()
    f = o.FIELDS[field]
    a = synth.ntt_input(log_n)
    w = np.array(f.limbs(f.omega(log_n)), np.uint64)
    reps, t = 0, time.perf_counter()
    while True:
        last = cpu_ref.best_fft(field, a, w, log_n, threads)
        reps += 1
        dt = time.perf_counter() - t
        if dt >= 5.0 or reps >= 16:
            break
    return ({"value": (1 << log_n) * reps / dt, "unit": "elems/s", "cores": threads, "kind": "port",
             "sample": f"{reps} x 2^{log_n} Fp best_fft (oracle/cpu_ref.cpp, {threads} threads), {dt:.2f} s"}, last)


TRAFFIC_SOURCE = ("profiles/traffic.json: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from committed rocprofv3 --pmc passes, one counter per pass "
                  "(headline sizes: tools/profile.sh over this same command; sweep sizes: tools/pmc_sweep.sh -> profiles/r05_pmc_sweep.md).  Factor 2 on FETCH_SIZE: "
                  "every fabric read request is a 128-byte line tallied at 64 bytes -- MI355X_MICROARCH.md's streaming calibration, measured to hold for this library's "
                  "16-byte-per-lane gathers out of random 128-byte records as well (profiles/r05_gather_calibration.txt: gather80 whole-lines / reported = 1.994, "
                  "TCC_EA0_RDREQ_32B = 0); WRITE_SIZE as reported.  Read from the file, NOT measured in this run")


_TRAFFIC_META = {}


def load_traffic(name: str):
    """PMC-derived HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/), or None.  An entry is either a number (rounds
    1 - 5) or {"bytes", "build", "kernel", "vgpr", "isa_instructions"}: the library build and the kernel's ISA the counters were collected on."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            ent = json.load(fh).get(name)
    except Exception:
        return None
    if isinstance(ent, dict):
        _TRAFFIC_META[name] = {k: v for k, v in ent.items() if k != "bytes"}
        return ent.get("bytes")
    if ent is not None:
        _TRAFFIC_META[name] = {"build": None}
    return ent


def traffic_provenance(name: str, version: str):
    """what the traffic figure of `name` was measured on, and whether that is this library: `traffic_stale` is True when the entry carries no
    build id or another one than trh_version() -- the kernels may have changed since the counters were collected"""
    meta = dict(_TRAFFIC_META.get(name) or {})
    m = __import__("re").search(r"build ([0-9a-f]+)", version or "")
    cur = m.group(1) if m else None
    meta["library_build"] = cur
    meta["traffic_stale"] = not (meta.get("build") and cur and meta["build"] == cur)
    return meta


Q_MOD = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001  # Pallas scalar field
P_MOD = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001  # Pallas base field = NTT field
FP_ROOT = 0x2BCE74DEAC30EBDA362120830561F81AEA322BF2B7BB7584BDAD6FABD87EA32F


def msm_roofline(n, acc_ms, tm, traffic):
    """the dominant kernel against the tier's roofline (HBM, algorithmic 96 B per pair) and against what really limits it
    (VALU issue: measured instruction mix x measured issue rates)"""
    ach = 96.0 * n / (acc_ms * 1e-3) / 1e9
    madds = n * min(int(tm["windows"]), -(-254 // int(tm["window_bits"])))  # the carry window above ceil(254 / c) is (almost surely) empty
    roof = {"bound": "hbm", "kernel": "msm_accumulate_seg_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "kernel_ms": acc_ms, "algorithmic_bytes": 96 * n,
            # the kernel is limited by VALU issue, not by HBM (see `valu`): the counter traffic is what it actually pulls per launch
            "limiter": "valu-issue", "traffic_gbs": (traffic / (acc_ms * 1e-3) / 1e9) if traffic else None}
    # issue model, corrected in round 4: every instruction class of the mixed addition is priced at its MARGINAL cost in a
    # stream of multiply-adds at the kernel's 3 waves per SIMD (tools/issue_probe.hip, profiles/issue_probe_r04.txt, normalised to
    # v_mad_i64_i32 = 4.00 cycles) -- round 3 had read the AVERAGE of an 8 mad + 4 add loop (3.6) as the cost of the add.  The counts are the
    # static instruction mix of the kernel's main line (llvm disassembly of msm_accumulate_seg_kernel<Fp>: blocks BB11_16 + BB11_26).
    mad_peak = 32.7e12  # lane-operations/s of a pure v_mad_i64_i32 stream (tools/microbench.hip)
    mad_equiv = sum(cnt * ISSUE_CYCLES[k] / 4.0 for k, cnt in MADD_MIX.items())
    other = sum(cnt for k, cnt in MADD_MIX.items() if k != "v_mad_i64_i32")
    t = acc_ms * 1e-3
    valu = {"mixed_adds_per_launch": madds, "mixed_adds_per_s": madds / t,
            "instruction_mix_per_mixed_add": MADD_MIX, "marginal_issue_cycles_at_3_waves_per_simd": ISSUE_CYCLES,
            "mad_i64_i32_per_s": MADD_MIX["v_mad_i64_i32"] * madds / t, "mad_peak_per_s": mad_peak, "other_valu_per_s": other * madds / t,
            "mad_equivalents_per_mixed_add": mad_equiv, "model_cycles_per_valu_instruction": 4.0 * mad_equiv / sum(MADD_MIX.values()),
            # the launch against the per-class issue model at the pure-multiply-add stream's rate (that stream holds ~2.0 GHz, this kernel ~1.82:
            # at the kernel's own clock -- SQ counters, profiles/r04_msm_stall_counters.txt -- the same model gives 3.69 / 4.06 = 0.91)
            "issue_frac": mad_equiv * madds / mad_peak / t,
            "issue_frac_all_non_mad_at_2_cycles": (MADD_MIX["v_mad_i64_i32"] + other * 0.5) * madds / mad_peak / t,
            "not_a_term": "instruction fetch (SQC_ICACHE hit rate 99.9999 %), exposed gather latency (7 % of wave time parked, overlapped by the SIMD's other two waves)"}
    return roof, valu


# instruction mix of one mixed addition in the signed 29-bit domain: 1733 VALU instructions per wave and mixed add (SQ_INSTS_VALU,
# profiles/*_msm_sq_counters.txt); by class from the disassembly of the kernel's main line: 8 x 81 + 2 x 45 product, 9 x 45 reduction and
# 3 x 9 subtrahend multiply-adds (+ 9 from address / bookkeeping arithmetic), the carry handling's mask / 64-bit shift / 64-bit add triples
MADD_MIX = {"v_mad_i64_i32": 1179, "v_and_b32": 156, "v_lshl_add_u64": 147, "v_ashrrev_i64": 144, "v_mov_b32": 23, "other_32bit_alu": 84}
# marginal SIMD cycles next to multiply-adds at 3 waves per SIMD (profiles/issue_probe_r04.txt, `norm` column; other_32bit_alu priced as v_add_u32)
ISSUE_CYCLES = {"v_mad_i64_i32": 4.0, "v_and_b32": 2.80, "v_lshl_add_u64": 3.58, "v_ashrrev_i64": 3.11, "v_mov_b32": 1.50, "other_32bit_alu": 2.80}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes -- this process has made no GPU call
    (not even `import torch`) and never will -- with the environment torch.distributed.run would give them, relay rank 0's JSON line,
    return the worst exit code.  If no rank printed a line (a rank died before the collective came up), print one that says so."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TRH_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + float(os.environ.get("TRH_BENCH_SPAWN_TIMEOUT", "1500"))
    line, first_fail = None, None
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    open_out = True
    while any(p.poll() is None for p in procs) or open_out:
        if open_out and sel.select(timeout=0.5):
            ln = procs[0].stdout.readline()
            if ln == "":
                open_out = False
                sel.unregister(procs[0].stdout)
            elif ln.startswith("{"):
                line = ln.strip()
        elif not open_out:
            time.sleep(0.2)
        rcs = [p.poll() for p in procs]
        if first_fail is None and any(rc not in (None, 0) for rc in rcs):
            first_fail = time.time()
        # a rank that died leaves the others in a rendezvous or a collective: give them a minute, then end OUR children by pid
        if (first_fail is not None and time.time() - first_fail > float(os.environ.get("TRH_BENCH_SPAWN_GRACE", "60"))) or time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    rcs = [p.wait() for p in procs]
    if line is None:
        line = json.dumps({"metric": METRIC, "value": None, "unit": "pairs/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                           "collective": {"backend": os.environ.get("TRH_BENCH_BACKEND", "nccl"), "world": args.gpus, "devices": None, "ok": False,
                                          "error": f"no rank printed a result line; exit codes {rcs}"}})
    print(line)
    sys.stdout.flush()
    return max([abs(rc) for rc in rcs] + [0]) and 1


def e2e_replays(modes=("resident", "dropin-batched", "dropin")):
    """BASELINE config 4 (k = 18, WORD_BITS = 32; /root/reference/src/test_utils.rs:41-49): the create_proof schedule replay over
    witness-shaped columns in each mode, the first results of the MSM / FFT / domain primitives of every run compared with the oracle
    (cpu_ref: checker only, outside every timed figure -- the replays time their steps with events / around their own calls)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ref
    from tiny_ram_halo2_amd import replay
    th = usable_cores()
    out = {"workload": "create_proof schedule of TinyRamCircuit<32, 8> (k = 18: 504 MSMs of 2^18 + 1 pairs, 497 iNTTs, 497 extended-coset transforms, lookups, "
                       "grand products, h(X), multiopen + IPA), witness-shaped columns, one proof per mode", "modes": {}}
    for mode in modes:
        stat = {"checked": 0, "failed": []}

        def hook(kind, inp, res, stat=stat):
            seen = stat.setdefault("seen", {})
            seen[kind] = seen.get(kind, 0) + 1
            first = seen[kind] == 1 or (kind == "commit_lagrange" and "column_class" in inp)
            ok = None
            if kind in ("commit_lagrange", "commit") and first:
                want = cpu_ref.to_affine("vesta", cpu_ref.best_multiexp("vesta", inp["scalars"], inp["bases"].download(), threads=th))
                ok = bool((np.asarray(res)[:8] == want).all())
            elif kind == "best_multiexp" and first:
                want = cpu_ref.to_affine(inp["curve"], cpu_ref.best_multiexp(inp["curve"], inp["scalars"], inp["bases"], threads=th))
                ok = bool((np.asarray(res)[:8] == want).all())
            elif kind == "best_fft" and first:
                ok = bool((np.asarray(res) == cpu_ref.best_fft(inp["field"], inp["a"], inp["omega"], inp["log_n"], threads=th)).all())
            elif kind == "best_fft_padded" and first:
                full = np.zeros((1 << inp["log_n"], 4), dtype=np.uint64)
                full[: inp["a"].shape[0]] = inp["a"]
                ok = bool((np.asarray(res) == cpu_ref.best_fft(inp["field"], full, inp["omega"], inp["log_n"], threads=th)).all())
            elif kind == "evals" and first:
                ok = bool((np.asarray(res) == cpu_ref.eval_polynomial(inp["field"], np.asarray(inp["a"]).reshape(-1, 4), inp["x"])).all())
            elif kind in ("lagrange_to_coeff", "coeff_to_extended", "coeff_to_extended_blocks") and first:
                field, j, k = inp["domain"]
                dom = cpu_ref.EvaluationDomain(field, j, k)
                a = np.asarray(inp["a"]).reshape(-1, 4)
                if kind == "lagrange_to_coeff":
                    ok = bool((np.asarray(res).reshape(-1, 4) == dom.lagrange_to_coeff(a)).all())
                else:
                    want = np.asarray(dom.coeff_to_extended(a)).reshape(-1, 4)
                    if kind == "coeff_to_extended":
                        ok = bool((np.asarray(res).reshape(-1, 4) == want).all())
                    else:  # block r, entry q == coeff_to_extended(a)[8 q + r]
                        nb, step = inp["n_blocks"], want.shape[0] // a.shape[0]
                        got = np.asarray(res).reshape(nb, a.shape[0], 4)
                        ok = all(bool((got[r] == want[r::step]).all()) for r in range(nb))
            if ok is not None:
                stat["checked"] += 1
                if not ok:
                    stat["failed"].append(kind)

        t0 = time.perf_counter()
        try:
            if mode == "resident":
                # several proofs in this process, as the reference proves sequentially in one (src/test_utils.rs:37-54): the first is reported beside the
                # steady state (the setup entries -- trh_bases_precompute / trh_bases_reserve / trh_domain_create / trh_domain_reserve -- build the
                # tables and size the scratch, so that no step of the first proof allocates)
                # (three replays: the process's FIRST proof untouched by checks -- a hook synchronises and computes on the host between the steps, the
                #  GPU idles and clocks down, and the steps after it run slower, which round 5's line counted as "first proof" cost --, then one that
                #  carries the oracle checks and is not timed for the summary, then the steady state)
                r1 = replay.run(32, batch=64, hook=None, verbose=False, columns="witness", keygen=False, gates_dir=os.path.join(ROOT, "tests", "golden"))
                replay.run(32, batch=64, hook=hook, verbose=False, columns="witness", keygen=False, gates_dir=os.path.join(ROOT, "tests", "golden"))
                r = replay.run(32, batch=64, hook=None, verbose=False, columns="witness", keygen=False, overlap=True, gates_dir=os.path.join(ROOT, "tests", "golden"))
                ent = {"gpu_ms_total": r["gpu_ms_total"], "gpu_ms_total_with_real_gates": r["gpu_ms_total_with_real_gates"], "gpu_ms": r["gpu_ms"], "extended_domain": r["extended_domain"],
                       # the per-column phase again with the transforms of batch i - 1 on a second libtrh context / stream / host thread under the
                       # commitments of batch i (what a restructured prover does; the step-by-step total above is what the two-function drop-in gets)
                       "column_loop_ms": r["column_loop_ms"], "gpu_ms_total_two_contexts": r["gpu_ms_total_two_contexts"],
                       "first_proof_in_process": {"gpu_ms_total": r1["gpu_ms_total"], "gpu_ms": r1["gpu_ms"]}, "scope": r["scope"]}
            else:
                r = replay.run_dropin(32, {"dropin": "literal", "dropin-batched": "batched"}[mode], batch=64, hook=hook, verbose=False, columns="witness")
                pc = r["pcie"]
                # one direction at a time at the probed pinned rate: for the bytes the caller handed over, and for the bytes that really crossed
                # (zero slots of the zero-padded vectors are cleared on the device)
                floor_ms = (pc["h2d_GB"] + pc["d2h_GB"]) / pc["link_peak_GBps_per_direction"] * 1e3
                cross_ms = (pc["h2d_GB"] - pc.get("h2d_zero_elided_GB", 0.0) + pc["d2h_GB"]) / pc["link_peak_GBps_per_direction"] * 1e3
                ent = {"wall_ms_incl_pcie_total": r["wall_ms_incl_pcie_total"], "wall_ms_incl_pcie": r["wall_ms_incl_pcie"], "pcie": pc,
                       "link_floor_ms": round(floor_ms, 1), "link_floor_frac": round(floor_ms / r["wall_ms_incl_pcie_total"], 3),
                       "link_floor_ms_bytes_that_crossed": round(cross_ms, 1), "link_frac_bytes_that_crossed": round(cross_ms / r["wall_ms_incl_pcie_total"], 3), "scope": r["scope"]}
            ent["checked_against_oracle"] = stat["checked"]
            ent["check"] = "oracle limb-for-limb ok" if not stat["failed"] and stat["checked"] else ("MISMATCH: " + ",".join(stat["failed"]) if stat["failed"] else "nothing checked")
        except Exception as exc:  # reported in the line, and a failed e2e fails the run
            ent = {"error": repr(exc)[:400], "check": "ERROR"}
        ent["seconds_incl_host_input_generation_and_checks"] = round(time.perf_counter() - t0, 2)
        out["modes"][mode] = ent
    return out


def native_replay(timeout_s: float = 600.0):
    """the compiled driver (examples/replay, C++ over include/trh.hpp: no Python between the steps) on the same k = 18 witness-shaped schedule,
    as a child process; its clock is the one quoted for the resident proof"""
    import subprocess
    exe = os.path.join(ROOT, "examples", "replay")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "examples/replay"], cwd=ROOT, capture_output=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "tiny-ram-halo2_amd") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run([exe, "--word-bits", "32", "--columns", "witness", "--overlap"], capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
        return json.loads(lines[-1])
    except Exception as exc:
        return {"error": repr(exc)[:300]}


def e2e_summary(e2e, native, ntt=None, sweep=None, scaling=None):
    """the proof-level totals as the LAST key of the line (the driver keeps the tail of stdout): one number per way of using the library,
    the secondary metric (BASELINE's metric names the 2^22 NTT) and BASELINE config 2 (the 2^20 MSM)"""
    m = e2e["modes"] if e2e else {}
    res, bat, lit = m.get("resident", {}), m.get("dropin-batched", {}), m.get("dropin", {})
    nat_ok = bool(native) and "error" not in native
    out = {
        "config": "k = 18 TinyRamCircuit<32, 8> schedule, witness-shaped columns, one MI355X",
        "resident_gpu_ms": round(native["ms_total"], 2) if nat_ok else res.get("gpu_ms_total"),
        "resident_clock": "examples/replay (compiled driver)" if nat_ok else "replay.py (Python mirror)",
        "resident_gpu_ms_python_mirror": res.get("gpu_ms_total"),
        "resident_two_ctx_gpu_ms": res.get("gpu_ms_total_two_contexts"),
        "resident_ipa_ms": (native.get("ms") or {}).get("ipa") if nat_ok else (res.get("gpu_ms") or {}).get("ipa"),
        "batched_wall_ms": bat.get("wall_ms_incl_pcie_total"),
        "literal_wall_ms": lit.get("wall_ms_incl_pcie_total"),
        "literal_commit_lagrange_ms": (lit.get("wall_ms_incl_pcie") or {}).get("commit_lagrange"),
        "literal_d2h_GBps_in_copies": (lit.get("pcie") or {}).get("d2h_GBps_in_copies"),
        "setup_tables_ms": native.get("setup_ms") if nat_ok else None,
        "setup_tables_GB": native.get("setup_tables_GB") if nat_ok else None,
        "keygen_gpu_ms": native.get("keygen_ms") if nat_ok else None,
        "native_checks_failed": native.get("checks_failed") if nat_ok else None,
        "checked_against_oracle": sum(int(v.get("checked_against_oracle", 0)) for v in m.values()),
        "check": "ok" if m and all(str(v.get("check", "")).startswith("oracle limb-for-limb ok") for v in m.values()) and (not nat_ok or native.get("checks_failed") == 0) else "see e2e",
    }
    first = (res.get("first_proof_in_process") or {}).get("gpu_ms_total")
    if first and res.get("gpu_ms_total"):
        out["first_proof_over_second"] = round(first / res["gpu_ms_total"], 4)
    if ntt:  # (the size is in the key: BASELINE's metric names 2^22, a test run may ask for another)
        out[f"ntt_2_{ntt.get('log_n', 22)}_elems_per_s"] = ntt.get("value")
        out[f"ntt_2_{ntt.get('log_n', 22)}_ms"] = ntt.get("ms_per_transform")
        out["ntt_frac"] = (ntt.get("roofline") or {}).get("frac")
    for ent in sweep or []:
        if ent.get("op") == "msm" and ent.get("log_n") == 20:
            out["msm_2_20_pairs_per_s"] = ent.get("value")
            out["msm_2_20_ms"] = ent.get("ms")
    if scaling:
        out["e2e_scaling"] = scaling
    if native and "error" in native:
        out["native_error"] = native["error"]
    return out


def single_process_child(devices, steps: int, log_n: int = 26):
    """ONE 2^26 Pallas MSM through trh_init_multi over `devices` from this process alone (what a single Rust prover process linking
    libtrh.so gets): bases range-sharded by the library, scalars resident on the first device and handed to the others with peer
    copies, partial points added on the host.  Prints one JSON line."""
    import torch
    from tiny_ram_halo2_amd import api, synth
    curve = "pallas"
    torch.cuda.set_device(devices[0])
    dev = torch.device("cuda", devices[0])
    api.init_multi(devices)
    api.set_shard_min(1 << 20)
    stream = torch.cuda.current_stream().cuda_stream
    n, block = 1 << log_n, 1 << 22
    reps = n // block
    bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    assert bases.shards() == len(devices), (bases.shards(), devices)
    blk = torch.from_numpy(synth.field_elements(synth.SEED_MSM | 22, block).view(np.int64)).to(dev)
    d_sc = blk.repeat(reps, 1).contiguous()
    for _ in range(2):
        res = bases.msm_dev(d_sc, n, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = bases.msm_dev(d_sc, n, stream=stream)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d_can = torch.empty_like(blk)
    api._check(api.lib().trh_field_op_dev(api.FQ, api.FIELD_OPS["from_mont"], api._devptr(blk), None, api._devptr(d_can), block, stream))
    torch.cuda.synchronize()
    can = d_can.cpu().numpy().view(np.uint64)
    t_0, t_1 = synth.weighted_scalar_sum(can, 1, 0), synth.weighted_scalar_sum(can, 0, 1)
    total = reps * (synth.BASE_S0 * t_0 + synth.BASE_D * t_1) + synth.BASE_D * block * t_0 * (reps * (reps - 1) // 2)
    api.set_shard_min(1 << 62)
    g = api.Bases.generate(curve, 1, 0, 1)
    want = g.msm(synth.ints_to_limbs([total % Q_MOD * ((1 << 256) % Q_MOD) % Q_MOD]))
    ok = bool((want == res).all())
    print(json.dumps({"workload": f"ONE 2^{log_n} Pallas MSM through trh_init_multi over {len(devices)} GPUs from one host process (a child of rank 0): bases range-sharded by "
                                  "the library, scalars resident on GPU 0 and handed to the other GPUs with peer copies (included), partial points copied to the host and added",
                      "value": n * steps / el, "unit": "pairs/s", "ms_per_msm": el / steps * 1e3, "check": "closed-form ok" if ok else "MISMATCH",
                      "peer_access": bool(api.lib().trh_group_peer_access())}))
    sys.stdout.flush()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--single-process-child", default=None, help="internal: comma-separated device list; run the single-process device-group MSM and exit")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=24, help="log2 of MSM pairs per GPU")
    ap.add_argument("--ntt-log-n", type=int, default=22)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the closed-form result check (keeps profiles free of the extra 1-pair MSM)")
    ap.add_argument("--global-log-n", type=int, default=0, help="strong scaling: 2^L pairs in TOTAL, range-sharded over the ranks (default: weak, 2^log-n per GPU)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the size sweep / strong-scaling / single-process extras (profiling runs)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the k = 18 create_proof replays (N = 1 only; implied by --no-sweep)")
    args = ap.parse_args()
    if args.single_process_child is not None:
        sys.exit(single_process_child([int(v) for v in args.single_process_child.split(",")], args.steps))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))  # before torch is imported: this process never touches a GPU

    import torch
    import torch.distributed as dist
    from tiny_ram_halo2_amd import api, sharded, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libtrh has no CPU fallback")
    # one rank per GPU; TRH_BENCH_BACKEND=gloo (+ ranks folded onto the GPUs present) exists only to
    # exercise the multi-rank code path on a 1-GPU box -- the driver's scaling runs use nccl (= RCCL)
    backend = os.environ.get("TRH_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else None
    props = torch.cuda.get_device_properties(dev_index)
    my_dev = f"cuda:{dev_index} {props.name} {getattr(props, 'gcnArchName', '')}".strip()
    try:  # the PCI address tells two physical devices apart (uuid / pci ids are not on every torch build)
        my_dev += f" pci {props.pci_bus_id:02x}:{props.pci_device_id:02x}"
    except Exception:
        pass
    collective = {"backend": None, "world": world, "devices": [my_dev], "ok": True}
    # under a launcher (WORLD_SIZE set) the process group comes up at EVERY world size: a one-rank RCCL group on one MI355X is the
    # first contact of this code with RCCL that a one-GPU box allows (tests/test_gpu_native.py::test_bench_one_rank_rccl_first_contact)
    # (a bare WORLD_SIZE=1 exported by a job scheduler, without a rendezvous address, is not a launcher: such a run benchmarks as one process)
    launched = "WORLD_SIZE" in os.environ and ("MASTER_PORT" in os.environ or "TORCHELASTIC_RUN_ID" in os.environ)
    if world > 1 or launched:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        collective = {"backend": backend, "world": world, "devices": None, "ok": False}
        try:
            tmo = datetime.timedelta(seconds=float(os.environ.get("TRH_BENCH_INIT_TIMEOUT", "180")))
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
            # proof that the backend really spans `world` ranks: every rank contributes 1 (on its GPU for RCCL) and names its device
            one = torch.ones(1, dtype=torch.int64, device=coll_dev)
            dist.all_reduce(one)
            names = [None] * world
            dist.all_gather_object(names, my_dev)
            collective = {"backend": dist.get_backend(), "world": dist.get_world_size(), "devices": names, "ok": int(one.item()) == world and dist.get_backend() == backend,
                          "all_reduce_of_ones": int(one.item())}
            if not collective["ok"]:
                raise RuntimeError(f"collective check failed: {collective}")
        except Exception as exc:  # never switch backend silently: the line says what failed and the run exits non-zero
            collective["error"] = repr(exc)[:400]
            if rank == 0:
                print(json.dumps({"metric": METRIC, "value": None, "unit": "pairs/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                                  "higher_is_better": True, "collective": collective}))
                sys.stdout.flush()
            sys.exit(1)
    api.init(dev_index)
    stream = torch.cuda.current_stream().cuda_stream
    curve = "pallas"
    failed = []

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def expected_point(total: int):
        """(total mod q) G through a one-pair MSM over the generator (bases are (s0 + i d) G with known logs)"""
        g = api.Bases.generate(curve, 1, 0, 1)
        return g.msm(synth.ints_to_limbs([total % Q_MOD * ((1 << 256) % Q_MOD) % Q_MOD]))

    def canonical(d_sc, count):
        d_can = torch.empty_like(d_sc)
        api._check(api.lib().trh_field_op_dev(api.FQ, api.FIELD_OPS["from_mont"], api._devptr(d_sc), None, api._devptr(d_can), count, stream))
        torch.cuda.synchronize()
        return d_can.cpu().numpy().view(np.uint64)

    class Workload:
        """this rank's slice [first, first + n) of a global MSM: resident bases (s0 + i d) G and Montgomery scalars.  Above 2^24 pairs
        per rank the scalars are a 2^22 block repeated (the host generator would take longer than the whole benchmark)"""

        def __init__(self, first, n):
            self.first, self.n = first, n
            self.bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n, first=first)
            if n <= (1 << 24):
                lg = max(n - 1, 1).bit_length()
                self.block, self.reps = n, 1
                sc_host = synth.field_elements(synth.SEED_MSM | lg, n, start=first)
                self.d_sc = torch.from_numpy(sc_host.view(np.int64)).to(dev)
            else:
                self.block = 1 << 22
                assert n % self.block == 0
                self.reps = n // self.block
                blk = torch.from_numpy(synth.field_elements(synth.SEED_MSM | 22, self.block).view(np.int64)).to(dev)
                self.d_sc = blk.repeat(self.reps, 1).contiguous()

        def local_msm(self):
            return self.bases.msm_dev(self.d_sc, self.n, stream=stream)

        def weighted_sum(self) -> int:
            """this slice's share of sum_i s_i (s0 + i d), exact"""
            can = canonical(self.d_sc[: self.block], self.block)
            if self.reps == 1:
                return synth.weighted_scalar_sum(can, synth.BASE_S0, synth.BASE_D, start=self.first)
            t0 = synth.weighted_scalar_sum(can, 1, 0)
            t1 = synth.weighted_scalar_sum(can, 0, 1)
            r = self.reps  # sum over repetitions of sum_i s_i (s0 + (first + rep * block + i) d)
            return r * ((synth.BASE_S0 + self.first * synth.BASE_D) * t0 + synth.BASE_D * t1) + synth.BASE_D * self.block * t0 * (r * (r - 1) // 2)

        def destroy(self):
            self.bases.destroy()
            del self.d_sc

    def time_msm(wl, steps, warmup, collective=True, events_outside=False):
        """`steps` timed passes (after `warmup`): local Pippenger -> one point; all-gather of the 96-byte partials + host add when
        `collective`; barrier + synchronize on both sides, MAX over ranks.  The kernel's HIP events are recorded inside the timed
        region (the headline); events_outside (the size sweep's small MSMs, where six event records per launch are 4 % of the step)
        times the steps without them first and collects the kernel durations in a second set of steps"""
        def step():
            if collective:
                return sharded.sharded_msm(curve, wl.local_msm, device=coll_dev)
            return wl.local_msm()
        result = None
        for _ in range(warmup):
            result = step()
        plain_elapsed = None
        if events_outside:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                result = step()
            torch.cuda.synchronize()
            plain_elapsed = time.perf_counter() - t0
        api.set_timing(True)
        acc_ms, phase = [], {"digits_ms": 0.0, "sort_ms": 0.0, "accumulate_ms": 0.0, "reduce_ms": 0.0, "total_ms": 0.0}
        fence() if collective else torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            result = step()
            tm = api.last_timing()
            acc_ms.append(tm["accumulate_kernel_ms"])
            for k in phase:
                phase[k] += tm[k] / steps
        fence() if collective else torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        api.set_timing(False)
        if collective:
            elapsed = max_over_ranks(elapsed)
        if plain_elapsed is not None:
            elapsed = plain_elapsed
        return result, elapsed, float(np.mean(acc_ms)), phase, api.last_timing()

    def check_msm(wl, result, collective=True):
        total = wl.weighted_sum()
        if collective and world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, total)
            total = sum(parts)
        if rank != 0:
            return None
        ok = bool((expected_point(total) == result).all())
        if not ok:
            failed.append("msm")
        return "closed-form ok" if ok else "MISMATCH"

    # ---- headline: weak scaling (2^log_n pairs per GPU) unless --global-log-n ------------------------------------------------
    if args.global_log_n:
        n_total = 1 << args.global_log_n
        lo, hi = sharded.shard_range(n_total, rank, world)
        mode = "strong"
    else:
        n_total = world << args.log_n
        lo, hi = rank << args.log_n, (rank + 1) << args.log_n
        mode = "weak"
    n = hi - lo
    wl = Workload(lo, n)
    result, elapsed, acc, phase, tm = time_msm(wl, args.steps, args.warmup)
    check = None if args.no_check else check_msm(wl, result)
    if world == 1 and launched:  # sharded_msm short-cuts a world of one: the gather of the 96-byte partial still goes through the backend once
        got = sharded.all_gather_points(np.asarray(result, dtype=np.uint64).reshape(12), device=coll_dev)
        collective["all_gather_points"] = "ok" if got.shape == (1, 12) and bool((got[0] == np.asarray(result, dtype=np.uint64).reshape(12)).all()) else "MISMATCH"
        if collective["all_gather_points"] != "ok":
            failed.append("all_gather_points")
    # the same step over a NOT-owned view of the same bases: libtrh then converts the 64-byte points to its 128-byte records inside
    # every MSM (msm_convert_bases_kernel) instead of once per resident handle -- the rate a caller without a long-lived handle gets
    with_conv = None
    if world == 1 and not args.no_sweep and n <= (1 << 25):
        view = api.Bases.wrap_device(curve, api.lib().trh_bases_device_ptr(wl.bases.handle), n)
        for _ in range(2):
            r2 = view.msm_dev(wl.d_sc, n, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(args.steps // 2, 3)):
            r2 = view.msm_dev(wl.d_sc, n, stream=stream)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / max(args.steps // 2, 3)
        with_conv = {"value": n / el, "unit": "pairs/s", "ms_per_step": el * 1e3, "same_point": bool((r2 == result).all()),
                     "what": "bases as plain 64-byte affine points in HBM, converted to the 128-byte records inside every MSM"}
        if not with_conv["same_point"]:
            failed.append("msm over the unconverted view")
        view.destroy()
    wl.destroy()

    # ---- secondary: Fp NTT @ 2^22 (same process, outside the MSM timed region) ----
    # every rank transforms its own column (create_proof's NTTs are independent per column: replicas, no collective);
    # the value is the whole-job rate over the slowest rank's time
    forward_kept = {}

    def time_ntt(ln, reps, warm, with_check, keep_forward=False):
        omega = pow(FP_ROOT, 1 << (32 - ln), P_MOD)
        mont = lambda v: synth.ints_to_limbs([v % P_MOD * ((1 << 256) % P_MOD) % P_MOD])[0]  # noqa: E731
        a = synth.ntt_input(ln)
        d_a = torch.from_numpy(a.view(np.int64).copy()).to(dev)
        chk = None
        if with_check:  # inverse(forward(a)) * n^-1 == a, and the forward transform is not the identity
            api.ntt_dev("fp", d_a, ln, mont(omega), stream=stream)
            torch.cuda.synchronize()
            if keep_forward:
                forward_kept[ln] = d_a.cpu().numpy().view(np.uint64).copy()
            moved = not bool((d_a[:4096].cpu().numpy().view(np.uint64) == a[:4096]).all())
            api.ntt_dev("fp", d_a, ln, mont(pow(omega, -1, P_MOD)), stream=stream)
            api.field_scale_dev("fp", d_a, 1 << ln, mont(pow(1 << ln, -1, P_MOD)), stream=stream)
            torch.cuda.synchronize()
            ok = moved and bool((d_a.cpu().numpy().view(np.uint64) == a).all())
            if not ok:
                failed.append(f"ntt 2^{ln}")
            chk = "inverse round trip ok" if ok else "MISMATCH"
        for _ in range(warm):
            api.ntt_dev("fp", d_a, ln, mont(omega), stream=stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        e0.record()
        for _ in range(reps):
            api.ntt_dev("fp", d_a, ln, mont(omega), stream=stream)
        e1.record()
        torch.cuda.synchronize()
        return max_over_ranks(e0.elapsed_time(e1) / reps), chk

    def ntt_entry(ln, ms, chk):
        ach = 64.0 * (1 << ln) / (ms * 1e-3) / 1e9
        traffic = load_traffic(f"ntt_fp_2^{ln}")
        return {"metric": f"Fp NTT elems/s @ 2^{ln}", "log_n": ln, "value": world * (1 << ln) / (ms * 1e-3), "unit": "elems/s", "ms_per_transform": ms, "check": chk,
                "roofline": {"bound": "hbm", "kernel": "ntt_passy_kernel (all passes of one transform)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": 64 << ln, "limiter": "valu-issue",
                             "traffic_gbs": (traffic / (ms * 1e-3) / 1e9) if traffic else None}}

    ln = args.ntt_log_n
    ms, chk = time_ntt(ln, max(args.steps, 5) * 4, max(args.warmup, 2), not args.no_check, keep_forward=world == 1 and not args.no_cpu_baseline)
    ntt = None
    if rank == 0:
        ntt = ntt_entry(ln, ms, chk)
        ntt["mode"] = "one transform per GPU at a time (independent columns, no collective)" if world > 1 else "single GPU"
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            ntt["cpu_baseline"], cpu_fwd = cpu_baseline_ntt("fp", ln)
            if ln in forward_kept:  # the oracle's transform of the same input, limb for limb (outside the timed region)
                ok = bool((forward_kept.pop(ln) == cpu_fwd).all())
                if not ok:
                    failed.append(f"ntt 2^{ln} vs oracle")
                ntt["check"] = (f"oracle limb-for-limb ok (cpu_ref.best_fft, 2^{ln}) + " + str(chk)) if ok else "MISMATCH vs oracle"
            del cpu_fwd

    # ---- extras outside the timed headline region ---------------------------------------------------------------------------
    sweep, strong, single = None, None, None
    if not args.no_sweep and world == 1:
        sweep = []
        for lg in (20, 22, 26):
            w2 = Workload(0, 1 << lg)
            steps = 6 if lg < 26 else 3
            res2, el2, acc2, ph2, tm2 = time_msm(w2, steps, 2, collective=False, events_outside=True)
            chk2 = check_msm(w2, res2, collective=False)
            w2.destroy()
            # beyond 2^25 pairs an MSM runs as range tiles of <= 2^25: the kernel time below is one tile's launch
            per_launch = (1 << lg) if lg <= 25 else (1 << 25)
            roof2, valu2 = msm_roofline(per_launch, acc2, tm2, load_traffic(f"msm_accumulate_2^{lg}"))
            sweep.append({"op": "msm", "curve": curve, "log_n": lg, "value": (1 << lg) * steps / el2, "unit": "pairs/s", "ms": el2 / steps * 1e3, "check": chk2,
                          "window_bits": tm2["window_bits"], "pairs_per_launch": per_launch, "roofline": roof2, "valu_issue_frac": valu2["issue_frac"],
                          "scalars": "uniformly random" if lg <= 24 else f"a uniformly random 2^22 block repeated {1 << (lg - 22)} times (the host generator would outlast the benchmark); bases all distinct"})
        for lg in (20, 24):
            ms2, chk2 = time_ntt(lg, 20, 3, True)
            e = ntt_entry(lg, ms2, chk2)
            sweep.append({"op": "ntt", "field": "fp", "log_n": lg, "value": e["value"], "unit": "elems/s", "ms": ms2, "check": chk2, "roofline": e["roofline"]})
    e2e = None
    if world == 1 and not args.no_sweep and not args.no_e2e:
        torch.cuda.empty_cache()
        e2e = e2e_replays()
        for mname, ent in e2e["modes"].items():  # a result that differs from the oracle's fails the run; a replay that could not run is reported in the line
            if str(ent.get("check", "")).startswith("MISMATCH"):
                failed.append(f"e2e {mname}: {ent.get('check')}")
        torch.cuda.empty_cache()
    if not args.no_sweep and world > 1 and mode == "weak":
        # BASELINE config 5: ONE 2^26 MSM over the N ranks (strong scaling), the same path as the headline
        L = 26
        lo5, hi5 = sharded.shard_range(1 << L, rank, world)
        w5 = Workload(lo5, hi5 - lo5)
        res5, el5, acc5, ph5, tm5 = time_msm(w5, max(args.steps // 2, 3), 2)
        chk5 = check_msm(w5, res5)
        w5.destroy()
        if rank == 0:
            strong = {"workload": f"ONE 2^{L} Pallas MSM range-sharded over {world} ranks ({hi5 - lo5} pairs each), {collective['backend']} all-gather of the 96-byte partials, host add",
                      "value": (1 << L) * max(args.steps // 2, 3) / el5, "unit": "pairs/s", "ms_per_msm": el5 / max(args.steps // 2, 3) * 1e3, "scaling": "strong", "check": chk5,
                      "accumulate_kernel_ms": acc5, "window_bits": tm5["window_bits"]}
        # the same MSM from ONE process through the C ABI's device group: a CHILD process of rank 0 (its own libtrh, trh_init_multi over the
        # N GPUs) under a timeout, so that a fault of this never-on-hardware-tested leg cannot take the scaling run with it; the other ranks
        # wait on the store, no GPU spinning
        if os.environ.get("TRH_BENCH_SINGLE_PROCESS", "1") != "0":
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                import subprocess
                devs = list(range(world)) if backend == "nccl" else [dev_index] * world  # gloo runs fold the ranks onto the GPUs present
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--single-process-child", ",".join(map(str, devs)), "--steps", str(max(args.steps // 2, 3))],
                                       capture_output=True, text=True, timeout=float(os.environ.get("TRH_BENCH_CHILD_TIMEOUT", "240")))
                    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    single = json.loads(lines[-1]) if lines else {"error": f"child exited {r.returncode}: {r.stderr[-300:]}"}
                    if single.get("check") == "MISMATCH":
                        failed.append("single-process msm")
                except subprocess.TimeoutExpired:
                    single = {"error": "timeout"}
                except Exception as exc:  # reported, never fatal to the headline
                    single = {"error": repr(exc)[:300]}
                store.set("trh_single_process_done", "1")
            else:
                import datetime
                store.wait(["trh_single_process_done"], datetime.timedelta(seconds=900))  # longer than the child's own timeout

    # ---- N > 1: one k = 18 resident proof PER GPU (the reference's multi-item workload is N independent proofs, /root/reference/src/test_utils.rs:37-54:
    #      inputs.iter().map(create_proof)); replicas, no data-path collective -- only the MAX over ranks of the proof's GPU time.  Per rank:
    #      ~25 GB of resident columns + 1.6 GB of fixed-base tables
    e2e_scaling = None
    if world > 1 and not args.no_sweep and not args.no_e2e:
        torch.cuda.empty_cache()
        mine, wall, err = -1.0, -1.0, None
        fence()  # (outside the try: every rank reaches it whatever happens to its own replay)
        try:
            from tiny_ram_halo2_amd import replay
            t0 = time.perf_counter()
            r = replay.run(32, batch=64, hook=None, device=dev_index, verbose=False, columns="witness", keygen=False, gates_dir=os.path.join(ROOT, "tests", "golden"))
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            mine = float(r["gpu_ms_total"])
        except Exception as exc:  # every rank still takes part in the reductions below
            err = repr(exc)[:300]
        ok_all = -max_over_ranks(-(1.0 if mine > 0 else 0.0))  # MIN over ranks
        slow = max_over_ranks(mine)
        slow_wall = max_over_ranks(wall)
        torch.cuda.empty_cache()
        if rank == 0:
            if ok_all > 0:
                e2e_scaling = {"proofs_per_s": world / (slow * 1e-3), "ms_per_proof_slowest_rank": slow, "n_gpus": world,
                               "mode": "one k = 18 TinyRamCircuit<32, 8> schedule replay per GPU, polynomials resident (replicas: no collective but the MAX over ranks)",
                               "clock": "replay.py (Python mirror): HIP events around every step of the proof, summed",
                               "wall_ms_slowest_rank_incl_host_input_generation": slow_wall, "ms_this_rank": mine}
            else:
                e2e_scaling = {"error": err or "a rank failed its replay", "n_gpus": world}

    if rank == 0:
        roof, valu = msm_roofline(n if n <= (1 << 25) else (1 << 25), acc, tm, load_traffic(f"msm_accumulate_2^{max(n - 1, 1).bit_length()}"))
        shape = f"2^{args.log_n} pairs per GPU; global size {world}*2^{args.log_n}" if mode == "weak" else f"2^{args.global_log_n} pairs in total, {n} per GPU"
        out = {
            "metric": METRIC,
            "value": n_total * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": mode,
            "vs_baseline": None,
            "dtype": "u255 (255-bit modular integers as 9 x 29/30-bit limbs in i32 / i64 lanes)",
            "data": "synthetic",
            "config": {"workload": f"Pallas MSM, {shape}, random 254-bit scalars, distinct bases (s0+i*d)G, inputs resident in HBM: scalars as 32-byte Montgomery words, "
                                   "bases in a resident handle, i.e. already converted to libtrh's 128-byte signed-limb records (once per handle, 1.0 ms at 2^24, "
                                   "outside the timed step; `with_base_conversion` is the rate with that conversion inside every step)", "curve": curve,
                       "pairs_per_gpu": n, "pairs_total": n_total, "mode": f"{mode} scaling" + (" (--global-log-n)" if mode == "strong" else " (default)"),
                       "window_bits": tm["window_bits"], "windows": tm["windows"],
                       "parallelism": f"range-shard x{collective['world']}, one process per GPU, {collective['backend']}" + (" (= RCCL)" if collective["backend"] == "nccl" else "") +
                                      " all-gather of 96-byte partials" if world > 1 else "single GPU"},
            "collective": collective,
            "roofline": roof,
            # what actually bounds the kernel: VALU issue.  Per mixed add 1170 v_mad_i64_i32 (8 products of 81, 2 squares of 45, 9 reductions
            # of 45, 27 for the fused subtrahends) and ~560 other ALU instructions (SQ_INSTS_VALU: 1733 per mixed add and wave); peaks are the
            # measured issue rates of tools/microbench.hip and tools/issue_probe.hip (32 x 32 multiply-add 32.7 T/s; other ALU instructions
            # 65 T/s alone, ~36 T/s between multiply-adds at this occupancy)
            "valu": valu,
            "phases_ms": phase,
            "check": check,
            "secondary": ntt,
        }
        out["roofline"]["traffic_source"] = TRAFFIC_SOURCE
        version = api.lib().trh_version().decode()
        out["library"] = version
        out["roofline"].update(traffic_provenance(f"msm_accumulate_2^{max(n - 1, 1).bit_length()}", version))
        if ntt is not None:
            ntt["roofline"]["traffic_source"] = TRAFFIC_SOURCE
            ntt["roofline"].update(traffic_provenance(f"ntt_fp_2^{ln}", version))
        for ent in sweep or []:
            ent["roofline"].update(traffic_provenance(("msm_accumulate_2^%d" if ent["op"] == "msm" else "ntt_fp_2^%d") % ent["log_n"], version))
        if sweep is not None:
            out["sweep"] = sweep
        if e2e is not None:
            out["e2e"] = e2e
        if strong is not None:
            out["strong"] = strong
        if single is not None:
            out["single_process"] = single
        if with_conv is not None:
            out["with_base_conversion"] = with_conv
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], cpu_log_n, cpu_point = cpu_baseline_msm(curve, args.log_n)
            # the oracle's point for the same inputs against the GPU's, limb for limb: the headline size when the CPU sample reached it
            # (it does on the driver's box), otherwise the largest size the sample ran, recomputed on the GPU for the comparison
            if cpu_log_n == args.log_n and mode == "weak":
                gpu_point = result
            else:
                wo = Workload(0, 1 << cpu_log_n)
                gpu_point = wo.local_msm()
                wo.destroy()
            ok = bool((np.asarray(gpu_point)[:8] == cpu_point).all())
            if not ok:
                failed.append(f"msm 2^{cpu_log_n} vs oracle")
            out["check"] = (f"oracle limb-for-limb ok (cpu_ref.best_multiexp, 2^{cpu_log_n}) + " + str(check)) if ok else "MISMATCH vs oracle"
        if e2e is not None:  # last key: the tail of the line is what the driver's record keeps
            out["e2e_summary"] = e2e_summary(e2e, native_replay(), ntt, sweep)
        elif e2e_scaling is not None:  # N > 1: one k = 18 proof per GPU (replicas), the proof-level scaling figure as the tail of the line
            out["e2e_summary"] = e2e_summary(None, None, ntt, None, e2e_scaling)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1 or launched:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.stderr.write("bench.py: result check FAILED: " + ", ".join(failed) + "\n")
        sys.exit(1)


if __name__ == "__main__":
    main()
