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
    # calibrate on 2^16 pairs, then size one MSM for <= ~12 s and repeat it until >= 10 s have been spent
    n0 = 1 << 16
    bases0 = cpu_ref.gen_bases(curve, synth.BASE_S0, synth.BASE_D, n0, threads)
    sc0 = synth.msm_scalars(16)
    t = time.perf_counter()
    cpu_ref.best_multiexp(curve, sc0, bases0, threads)
    rate0 = n0 / (time.perf_counter() - t)
    log_n = 16
    while log_n < log_cap and (1 << (log_n + 1)) / rate0 < 12.0:
        log_n += 1
    n = 1 << log_n
    bases = cpu_ref.gen_bases(curve, synth.BASE_S0, synth.BASE_D, n, threads)
    sc = synth.msm_scalars(log_n)
    reps, t = 0, time.perf_counter()
    while True:
        cpu_ref.best_multiexp(curve, sc, bases, threads)
        reps += 1
        dt = time.perf_counter() - t
        if dt >= 10.0 or reps >= 8:
            break
    return {"value": n * reps / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x 2^{log_n} Pallas best_multiexp (oracle/cpu_ref.cpp, {threads} threads), {dt:.2f} s"}


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
        cpu_ref.best_fft(field, a, w, log_n, threads)
        reps += 1
        dt = time.perf_counter() - t
        if dt >= 5.0 or reps >= 16:
            break
    return {"value": (1 << log_n) * reps / dt, "unit": "elems/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x 2^{log_n} Fp best_fft (oracle/cpu_ref.cpp, {threads} threads), {dt:.2f} s"}


def load_traffic(name: str):
    """PMC-derived HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(name)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=24, help="log2 of MSM pairs per GPU")
    ap.add_argument("--ntt-log-n", type=int, default=22)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the closed-form result check (keeps profiles free of the extra 1-pair MSM)")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    api.init(dev_index)
    stream = torch.cuda.current_stream().cuda_stream

    curve = "pallas"
    log_n = args.log_n
    n = 1 << log_n
    first = rank * n  # this rank's slice of the global (scalar, base) range
    bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n, first=first)
    sc_host = synth.field_elements(synth.SEED_MSM | log_n, n, start=first)
    d_sc = torch.from_numpy(sc_host.view(np.int64)).to(dev)

    def step():
        # local Pippenger -> one Jacobian point; all-gather of the 96-byte partials + host add when world > 1
        return sharded.sharded_msm(curve, lambda: bases.msm_dev(d_sc, n, stream=stream), device=coll_dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        result = step()
    api.set_timing(True)
    acc_ms, phase = [], {"digits_ms": 0.0, "sort_ms": 0.0, "accumulate_ms": 0.0, "reduce_ms": 0.0, "total_ms": 0.0}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
        tm = api.last_timing()
        acc_ms.append(tm["accumulate_kernel_ms"])
        for k in phase:
            phase[k] += tm[k] / args.steps
    fence()
    elapsed = time.perf_counter() - t0
    api.set_timing(False)
    tm = api.last_timing()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # closed-form check of the whole (global) MSM: bases are (s0 + i d) G with known logs
    check = None
    if not args.no_check:
        d_can = torch.empty_like(d_sc)
        api._check(api.lib().trh_field_op_dev(api.FQ, api.FIELD_OPS["from_mont"], api._devptr(d_sc), None, api._devptr(d_can), n, stream))
        torch.cuda.synchronize()
        can = d_can.cpu().numpy().view(np.uint64)
        q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
        total = synth.weighted_scalar_sum(can, synth.BASE_S0, synth.BASE_D, start=first) % q   # this rank's share of sum s_i (s0 + i d)
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, total)
            total = sum(parts) % q
        if rank == 0:
            R = (1 << 256) % q
            g = api.Bases.generate(curve, 1, 0, 1)  # the generator itself
            want = g.msm(synth.ints_to_limbs([total * R % q]))
            check = "closed-form ok" if (want == result).all() else "MISMATCH"

    # ---- secondary: Fp NTT @ 2^22 (same process, outside the MSM timed region) ----
    # every rank transforms its own column (create_proof's NTTs are independent per column: replicas, no collective);
    # the value is the whole-job rate over the slowest rank's time
    ntt = None
    ln = args.ntt_log_n
    P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
    root = 0x2BCE74DEAC30EBDA362120830561F81AEA322BF2B7BB7584BDAD6FABD87EA32F
    omega = pow(root, 1 << (32 - ln), P)
    w = synth.ints_to_limbs([omega * ((1 << 256) % P) % P])[0]
    a = synth.ntt_input(ln)
    d_a = torch.from_numpy(a.view(np.int64).copy()).to(dev)
    for _ in range(max(args.warmup, 2)):
        api.ntt_dev("fp", d_a, ln, w, stream=stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(args.steps, 5) * 4
    fence()
    e0.record()
    for _ in range(reps):
        api.ntt_dev("fp", d_a, ln, w, stream=stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if world > 1:
        t = torch.tensor([ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    if rank == 0:
        ach = 64.0 * (1 << ln) / (ms * 1e-3) / 1e9
        ntt = {"metric": f"Fp NTT elems/s @ 2^{ln}", "value": world * (1 << ln) / (ms * 1e-3), "unit": "elems/s", "ms_per_transform": ms,
               "mode": "one transform per GPU at a time (independent columns, no collective)" if world > 1 else "single GPU",
               "roofline": {"bound": "hbm", "kernel": "ntt_passz_kernel (x3 passes)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                            "traffic": load_traffic(f"ntt_fp_2^{ln}")}}
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            ntt["cpu_baseline"] = cpu_baseline_ntt("fp", ln)

    if rank == 0:
        acc = float(np.mean(acc_ms))
        # scalars are uniform below 2^254, so the carry window above ceil(254 / c) full windows is (almost surely) empty
        madds = n * min(int(tm["windows"]), -(-254 // int(tm["window_bits"])))
        ach = 96.0 * n / (acc * 1e-3) / 1e9
        out = {
            "metric": METRIC,
            "value": world * n * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"Pallas MSM, 2^{log_n} pairs per GPU, random 254-bit scalars, distinct bases (s0+i*d)G, "
                                   f"inputs resident in HBM; global size {world}*2^{log_n}", "curve": curve,
                       "pairs_per_gpu": n, "window_bits": tm["window_bits"], "windows": tm["windows"],
                       "parallelism": f"range-shard x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_seg_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": load_traffic(f"msm_accumulate_2^{log_n}"),
                         "kernel_ms": acc, "algorithmic_bytes": 96 * n},
            # what actually bounds the kernel: VALU issue.  Per mixed add 8 fz_mul (126 v_mad_u64_u32 each) + 2 fz_sqr (90) =
            # 1188 half-rate multiply-adds and ~1130 other ALU instructions (SQ_INSTS_VALU: 2317 per mixed add and wave); peaks are the
            # measured issue rates of tools/microbench.hip (profiles/microbench_r01m.txt: v_mad_u64_u32 32.7 T/s, v_add_u32 65 T/s)
            "valu": {"mixed_adds_per_launch": madds, "mixed_adds_per_s": madds / (acc * 1e-3),
                     "mad_u64_u32_per_s": 1188 * madds / (acc * 1e-3), "mad_peak_per_s": 32.7e12,
                     "other_valu_per_s": 1130 * madds / (acc * 1e-3), "other_peak_per_s": 65e12,
                     "issue_frac": (1188 * madds / 32.7e12 + 1130 * madds / 65e12) / (acc * 1e-3)},
            "phases_ms": phase,
            "check": check,
            "secondary": ntt,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_msm(curve)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
