"""Replay of the arithmetic schedule `halo2_proofs::plonk::create_proof` issues for the reference's
TinyRamCircuit<WORD_BITS, REG_COUNT = 8> (the reference call
site is /root/reference/src/test_utils.rs:41-49 with k = 2 + WORD_BITS / 2 from :20).

The Rust prover cannot run here (no toolchain), so this driver issues the same primitive kinds, sizes
and counts against libtrh with synthetic column data: per column a `commit_lagrange` (MSM of n + 1
pairs over the resident Lagrange bases), `lagrange_to_coeff` (iNTT n) and `coeff_to_extended` (coset NTT
8n); the permuted input / table columns of the 31 lookups and the 47 + 31 grand-product columns; the lookup / permutation / vanishing commitments; the extended iNTT of h(X); five h-piece commits;
and one IPA opening (k rounds).  Column / lookup / permutation counts are derived from the reference's
`configure` code (Appendix B): 94 instance + 263 advice columns, 31 lookups, 47 permutation products,
quotient degree 5 => extended_k = k + 3.  The h(X) numerator runs on the device as well (`expr.GateEvaluator`
over the resident extended cosets) -- with a SYNTHETIC gate set of the reference's shape (selector-gated
constraints up to degree 6), because the real one is the circuit definition and needs the Rust toolchain to
extract.  Witness generation and the transcript stay on the host and are not part of the replay.

    python -m tiny_ram_halo2_amd.replay --word-bits 32           # k = 18, the 2^14-cycle configuration
    python -m tiny_ram_halo2_amd.replay --word-bits 16           # k = 10 (BASELINE config 1's circuit size)
    python -m tiny_ram_halo2_amd.replay --columns witness        # columns shaped like the reference's witness (see witness_columns)

`--columns witness` replaces the uniformly random columns by the value classes the reference's tables really hold: flags and
<= WORD_BITS-bit words on the n / 4 live rows, zero padding behind them, a few blinding rows at the end, sorted small values
for the permuted lookup columns, and full-size field elements only for the grand-product columns -- which is what puts
almost every pair of a commitment into a handful of buckets (the skew paths of the MSM).  The `keygen` section replays
keygen_vk / keygen_pk (fixed and sigma columns: commit, iNTT, coset NTT; the l0 / l_blind / l_last cosets), once per proving
key; the `multiopen` entry runs the whole poly::multiopen::create_proof (x1 folds over every queried column, kate divisions,
x2 fold, q' commitment, evaluations, x4 fold, IPA) on the resident coefficient forms.

`run(..., hook=f)` calls f(kind, inputs, outputs) on the first items of every primitive kind so that a test
can compare them with the oracle (tests/test_gpu_replay.py); the module itself never touches the oracle.
"""
from __future__ import annotations

import argparse
import os
import sys
import json
import time

import numpy as np

from . import api, expr, ipa, permutation, poly, synth

# Appendix B counts for TinyRamCircuit<WB, 8>
N_INSTANCE, N_ADVICE, N_LOOKUPS, N_PERM_PRODUCTS, N_H_PIECES = 94, 263, 31, 47, 5
N_SYNTH_GATES = 300  # gate polynomials of the synthetic h(X) step (the real count is a property of the circuit definition)
# keygen: fixed columns (pc, time, the selectors, the out / even-bits / pow lookup tables, the constants column:
# /root/reference/src/circuits/tables/prog.rs:140-144, exe.rs:538-551, aux/out_table.rs:98-100, even_bits.rs:51, 330, pow.rs:16-17,
# assign.rs:25) and one sigma polynomial per equality-enabled column (prog.rs:151-152: 188)
N_FIXED, N_SIGMA = 25, 188
BLINDING_ROWS = 6  # cs.blinding_factors() + 1 rows at the end of every advice / permuted / product column hold random values
QUOTIENT_J = 6  # cs.degree() = 6 => EvaluationDomain::new(6, k): quotient_poly_degree 5, extended_k = k + 3


def column_classes(word_bits: int):
    """value classes of the 497 Lagrange-basis columns of one proof, in commitment order (instance, advice, permuted lookup
    columns, product columns): (count, kind, blinded).  Counted from the reference's tables: the exe table's 169 advice columns are
    the instruction-decoding flags of the 94-column program line, flag / selector bits of the temp-var machinery and the signed /
    logic / shift chips, and WORD_BITS-bit words (pc, registers, address, value, temp a..d, decompositions) with their even / odd
    halves as even-bits words (/root/reference/src/circuits/tables/exe.rs:538-741, exe/temp_vars.rs:42-119, even_bits.rs:90-107);
    the prog table mirrors the program line as instance + advice (tables/prog.rs:139-161).  29 of the 31 lookups compare ONE
    even-bits / shift expression with a range table (sorted small values), the CorrectOut and program lookups compress 15 / 94
    expressions with theta (full-size values)."""
    return [
        (70, "flag", False), (24, "word", False),                       # instance: the program (flags of the decoded line, immediates)
        (150, "flag", True), (90, "word", True), (23, "even", True),     # advice
        (58, "sorted", True), (4, "full", True),                         # permuted input / table columns of the lookups
        (N_LOOKUPS + N_PERM_PRODUCTS, "full", True),                     # grand-product columns z
    ]


def witness_columns(kind: str, blinded: bool, seed: int, b: int, n: int, word_bits: int) -> np.ndarray:
    """(b, n, 4) CANONICAL limbs of b columns of one class: live rows are the first n / 4 (exe.rs:106, prog.rs:137), zero behind
    them, BLINDING_ROWS random rows at the very end"""
    live = n // 4
    out = np.zeros((b, n, 4), dtype=np.uint64)
    raw = synth.splitmix64_stream(seed, 0, b * live).reshape(b, live)
    if kind == "flag":
        out[:, :live, 0] = raw & np.uint64(1)
    elif kind == "word":
        out[:, :live, 0] = raw & np.uint64((1 << word_bits) - 1)
    elif kind == "even":   # bits at even positions only (EvenBitsConfig decompositions)
        out[:, :live, 0] = raw & np.uint64(int("01" * (word_bits // 2), 2))
    elif kind == "sorted":  # permuted lookup column: the sorted multiset of a range-table lookup
        out[:, :live, 0] = np.sort(raw & np.uint64((1 << (word_bits // 2)) - 1), axis=1)
    else:
        out[:] = synth.field_elements(seed, b * n).reshape(b, n, 4)
    if blinded and kind != "full":
        out[:, n - BLINDING_ROWS:] = synth.field_elements(seed ^ 0xB11D, b * BLINDING_ROWS).reshape(b, BLINDING_ROWS, 4)
    return out


def schedule(k: int) -> dict:
    lag_cols = N_INSTANCE + N_ADVICE + 3 * N_LOOKUPS + N_PERM_PRODUCTS
    return {
        "k": k, "n": 1 << k, "extended_k": k + 3,
        "msm_n_plus_1": N_INSTANCE + N_ADVICE + 2 * N_LOOKUPS + N_LOOKUPS + N_PERM_PRODUCTS + 1 + N_H_PIECES + 1,
        "intt_n": lag_cols, "ntt_extended": lag_cols, "intt_extended": 1, "ipa_openings": 1,
    }


class _FixedTranscript:
    """challenge source standing in for the host's BLAKE2b transcript"""

    def __init__(self, m, seed=0x7E57):
        self.m, self.ctr, self.seed = m, 0, seed

    def write_point(self, p):
        pass

    def write_scalar(self, s):
        pass

    def squeeze_challenge_scalar(self):
        self.ctr += 1
        w = [int(v) for v in synth.splitmix64_stream(self.seed, 4 * self.ctr, 4)]
        return (w[0] | w[1] << 64 | w[2] << 128 | w[3] << 192) % self.m


def _warm_clocks(dom, n, dev, limit_s: float = 5.0) -> dict:
    """Shader clocks of an idle GPU take a while to come up (the first process on a fresh box ran its whole per-column phase 3.7 x
    slower, memory-bound steps unaffected): repeat a small batch of transforms until its time has settled near the usual figure, so that
    the timed phase that follows measures the arithmetic and not the ramp.  Returns what it saw (kept in the result line)."""
    import torch
    t = torch.zeros((16, n, 4), dtype=torch.int64, device=dev)
    seen = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < limit_s:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            dom.lagrange_to_coeff(t)
        e1.record()
        torch.cuda.synchronize()
        seen.append(e0.elapsed_time(e1) / 8)
        if len(seen) >= 6 and max(seen[-4:]) < 1.03 * min(seen[-4:]) and seen[-1] < 1.15 * min(seen):
            break
    return {"iterations": len(seen), "first_ms": round(seen[0], 3), "last_ms": round(seen[-1], 3), "seconds": round(time.perf_counter() - t0, 2)}


def _column_loop_two_contexts(params, dom, batches, blinds, ext_buf, D, blocks, x_eval, field, n, dev, overlapped: bool) -> float:
    """The per-column phase of create_proof (commit_lagrange, lagrange_to_coeff, coeff_to_extended, the evaluations) over resident
    column batches, either one step after the other or with the transforms of batch i - 1 on a SECOND libtrh context (own scratch, own
    stream, a second host thread) while the first commits batch i: the batched MSM's latency-bound sort / bucket-reduction tails fill
    with transform work.  Returns the wall time in ms (device synchronised on both sides)."""
    import threading

    import torch
    ctx2 = api.Context(dev.index)
    s2 = torch.cuda.Stream(device=dev)
    err = []

    def transforms(cols):
        try:
            ctx2.bind()
            with torch.cuda.stream(s2):
                coeff = dom.lagrange_to_coeff(cols)
                if blocks:
                    dom.coeff_to_extended_blocks(coeff, D, out=ext_buf)
                else:
                    dom.coeff_to_extended(coeff, out=ext_buf)
                api.poly_eval_batch_dev(field, coeff, n, cols.shape[0], x_eval, stream=s2.cuda_stream)
                s2.synchronize()
        except Exception as e:  # surfaced by the caller
            err.append(e)
        finally:
            api.Context.unbind()

    try:
        for w in range(min(3, len(batches))):  # untimed: the second context builds its twiddle tables, and the clocks are up when the
            warm = batches[w].clone()          # timed loop starts (the caller has just spent seconds generating columns on the host)
            params.commit_lagrange_batch(warm, blinds[w])
            transforms(warm)
            del warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prev = None
        for cols, bl in zip(batches, blinds):
            th = None
            if prev is not None and overlapped:
                th = threading.Thread(target=transforms, args=(prev,))
                th.start()
            elif prev is not None:
                ta = time.perf_counter()
                transforms(prev)
                if os.environ.get("TRH_REPLAY_VERBOSE"):
                    print(f"  transforms: {(time.perf_counter() - ta) * 1e3:.2f} ms", file=sys.stderr)
            ta = time.perf_counter()
            params.commit_lagrange_batch(cols, bl)
            if os.environ.get("TRH_REPLAY_VERBOSE"):
                torch.cuda.synchronize()
                print(f"  commit: {(time.perf_counter() - ta) * 1e3:.2f} ms", file=sys.stderr)
            if th is not None:
                th.join()
            prev = cols
        transforms(prev)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    finally:
        ctx2.destroy()
    if err:
        raise err[0]
    return ms


def run(word_bits: int, batch: int = 64, hook=None, device: int = 0, verbose: bool = True, precompute: bool = True, columns: str = "random",
        keygen: bool = True, extended: str = "blocks", overlap: bool = False, gates_dir: str | None = None) -> dict:
    import torch

    from . import multiopen
    assert columns in ("random", "witness") and extended in ("blocks", "full")
    # extended = "blocks": the extended domain as cosets of the size-2^k subgroup, only the QUOTIENT_J - 1 = 5 of 8 the quotient needs
    # (csrc/domain.hip); "full": EvaluationDomain::coeff_to_extended as halo2 0.2.0 has it (all 2^extended_k points, natural order)
    blocks = extended == "blocks"
    D = QUOTIENT_J - 1

    k = 2 + word_bits // 2
    sch = schedule(k)
    n, ek = sch["n"], sch["extended_k"]
    curve, field = "vesta", "fp"  # the reference proves over Fp with Params<EqAffine> (test_utils.rs:8, 21)
    api.init(device)
    dev = torch.device("cuda", device)
    dom = poly.EvaluationDomain(field, QUOTIENT_J, k)
    assert dom.extended_k == ek, (dom.extended_k, ek)

    # Params: synthetic resident bases (Params::new's hash-to-curve generators are a setup cost, a "next" row)
    g = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n + 1)
    gl = api.Bases.generate(curve, synth.BASE_S0 + 77, synth.BASE_D + 2, n + 1)
    g_host = g.download()
    params = poly.Params.__new__(poly.Params)
    params.curve, params.k, params.n = curve, k, n
    params._g, params._g_lagrange = g, gl
    params.w = g_host[n:n + 1]
    params.u = api.Bases.generate(curve, 4242, 1, 1).download()
    # keygen-time setup: fixed-base tables of the two base sets (reported separately; amortised over every proof made with the key)
    torch.cuda.synchronize()
    t_pre = time.perf_counter()
    if precompute:
        params.precompute()
        params.ipa_bases()  # the opening's resident set g || w || u with its own table
        params.reserve(batch)  # keygen-time sizing of the scratch: the first proof of the process allocates and builds nothing inside its steps
        dom.reserve(batch)
    torch.cuda.synchronize()
    precompute_ms = (time.perf_counter() - t_pre) * 1e3

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    times = {"lookup_permute": 0.0, "product_columns": 0.0, "commit_lagrange": 0.0, "lagrange_to_coeff": 0.0, "coeff_to_extended": 0.0, "evals": 0.0, "h_eval": 0.0, "commit": 0.0,
             "extended_to_coeff": 0.0, "multiopen_folds": 0.0, "ipa": 0.0}
    counts = {kk: 0 for kk in times}
    checked = 0
    torch.cuda.synchronize()
    t_wall = time.perf_counter()

    # --- lookup argument: permuted input / table columns of the 31 lookups (permute_expression_pair) ---
    distinct = torch.from_numpy(synth.field_elements(0x7AB1E, min(n, 1 << 16)).view(np.int64)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x100C)
    tabs = torch.stack([distinct[torch.arange(n, device=dev) % distinct.shape[0]] for _ in range(N_LOOKUPS)]).contiguous()
    inps = torch.stack([distinct[torch.randint(0, distinct.shape[0], (n,), device=dev, generator=gen)] for _ in range(N_LOOKUPS)]).contiguous()
    permutation.lookup_permute_batch(field, inps, tabs)              # untimed first call: the sort's scratch is allocated once per process
    e0 = ev()
    pas, pss = permutation.lookup_permute_batch(field, inps, tabs)   # the 31 lookups of the proof in one call
    e1 = ev()
    torch.cuda.synchronize()
    times["lookup_permute"] += e0.elapsed_time(e1)
    counts["lookup_permute"] += N_LOOKUPS
    if hook is not None:
        hook("lookup_permute", dict(input=inps[0].cpu().numpy().view(np.uint64), table=tabs[0].cpu().numpy().view(np.uint64), field=field),
             (pas[0].cpu().numpy().view(np.uint64), pss[0].cpu().numpy().view(np.uint64)))
        checked += 1
    del tabs, inps, pas, pss
    del distinct

    # --- grand products: the 47 permutation product columns (4 columns each) and the 31 lookup products, all at once ---
    m_ = poly._MODULUS[field]
    beta, gamma = 0xBE7A % m_, 0x6A33A % m_
    wit = [torch.from_numpy(synth.field_elements(0x9E0 + j, n).view(np.int64)).to(dev) for j in range(8)]
    om = torch.empty((n, 4), dtype=torch.int64, device=dev)
    api.powers_dev(field, om, n, expr._limbs(field, permutation.omega(field, k)))
    num_rows, den_rows = [], []
    for c in range(N_PERM_PRODUCTS):  # plonk/permutation/prover.rs: chunks of 4 columns, the same witness columns stand in for every chunk
        nr, dr = permutation.permutation_terms(field, wit[:4], wit[4:], om, beta, gamma, first_column=4 * c)
        num_rows.append(nr); den_rows.append(dr)
    for li in range(N_LOOKUPS):       # plonk/lookup/prover.rs commit_product
        nr, dr = permutation.lookup_terms(field, *[wit[(i + li) % 8] for i in range(4)], beta, gamma)
        num_rows.append(nr); den_rows.append(dr)
    permutation.grand_products_terms(field, k, num_rows, den_rows)  # untimed first call: the inversion's scratch is allocated once per process
    e0 = ev()
    zs = permutation.grand_products_terms(field, k, num_rows, den_rows)
    e1 = ev()
    torch.cuda.synchronize()
    times["product_columns"] += e0.elapsed_time(e1)
    counts["product_columns"] += len(num_rows)
    if hook is not None:
        hook("product_column", dict(values=[w.cpu().numpy().view(np.uint64) for w in wit[:4]], sigmas=[w.cpu().numpy().view(np.uint64) for w in wit[4:]],
                                    beta=beta, gamma=gamma, first_column=4, field=field, k=k), zs[1].cpu().numpy().view(np.uint64))
        checked += 1
    del zs, num_rows, den_rows, wit, om

    x_eval = synth.field_elements(0xE7A, 1)[0]
    ext_rows = D * n if blocks else 1 << ek
    ext_buf = torch.empty((min(batch, sch["intt_n"]), ext_rows, 4), dtype=torch.int64, device=dev)  # the batch's extended cosets, reused
    # --- Lagrange-basis columns: instance, advice, lookup permuted x2 + z, permutation z ---
    lag_total = sch["intt_n"]
    done = 0
    seed = 0xC01
    # the coefficient forms stay resident for the multiopen argument at the end (497 x 2^k x 32 B: 4.2 GB at k = 18)
    coeff_all = torch.empty((lag_total + 1 + N_H_PIECES, n, 4), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    if columns == "witness":  # (kind, blinded) of every column, in commitment order
        kinds = [(kind, blinded) for count, kind, blinded in column_classes(word_bits) for _ in range(count)]
        assert len(kinds) == lag_total

    def make_columns(first, b):
        """-> (host limbs (b, n, 4) Montgomery, device tensor).  Witness-shaped columns are built as small canonical integers and
        brought to the Montgomery form on the device (what the prover's field elements are in memory)"""
        if columns == "random":
            h = synth.field_elements(seed + first, b * n).reshape(b, n, 4)
            return h, torch.from_numpy(h.view(np.int64)).to(dev)
        can = np.empty((b, n, 4), dtype=np.uint64)
        i = 0
        while i < b:  # runs of one class
            j = i
            while j < b and kinds[first + j] == kinds[first + i]:
                j += 1
            can[i:j] = witness_columns(kinds[first + i][0], kinds[first + i][1], seed + first + i, j - i, n, word_bits)
            i = j
        d = torch.from_numpy(can.view(np.int64)).to(dev)
        api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS["to_mont"], api._devptr(d), None, api._devptr(d), b * n, stream))
        torch.cuda.synchronize()
        return (d.cpu().numpy().view(np.uint64).reshape(b, n, 4) if hook is not None and first == 0 else None), d

    # with a hook, one commitment of EVERY value class is handed over (the first column of each class): flags, words, even-bits words,
    # sorted lookup columns and full-size values, blinded and not, take different paths through the MSM (chunked bucket passes, heavy
    # buckets, the plain accumulation)
    class_first = {}
    if hook is not None and columns == "witness":
        for idx, kb in enumerate(kinds):
            class_first.setdefault(kb, idx)
    class_first = set(class_first.values())
    # the columns are built first (untimed: witness generation is the host's, and the GPU would otherwise idle between the batches and
    # run every timed burst on ramping clocks -- measured: 3.7 x slower commitments and transforms on a box with eager power management);
    # they wait as Lagrange forms in the rows of coeff_all, which the in-place transforms below turn into the coefficient forms
    cols_h0 = None
    while done < lag_total:
        b = min(batch, lag_total - done)
        cols_h, cols = make_columns(done, b)
        if done == 0:
            cols_h0 = cols_h
        coeff_all[done:done + b].copy_(cols)
        del cols
        done += b
    torch.cuda.synchronize()
    # untimed first call of the batched commitment (as for the lookups and the product columns above): the MSM's scratch for a batch of this
    # size is allocated once per process -- ~8 ms of hipMalloc inside the first timed batch otherwise; bench.py's e2e reports a process's first
    # proof beside its second for the same reason
    _wb = min(batch, lag_total)
    params.commit_lagrange_batch(coeff_all[:_wb].clone(), synth.field_elements(seed + 0x200000, _wb))
    torch.cuda.synchronize()
    warm = _warm_clocks(dom, n, dev)
    done = 0
    while done < lag_total:
        b = min(batch, lag_total - done)
        cols_h, cols = cols_h0, coeff_all[done:done + b]
        blinds = synth.field_elements(seed + 0x100000 + done, b)
        e0 = ev()
        pts = params.commit_lagrange_batch(cols, blinds)
        e1 = e1_commit = ev()
        for i in range(b):
            if done + i in class_first and not (done == 0 and i < 3):
                torch.cuda.synchronize()
                hook("commit_lagrange", dict(scalars=np.concatenate([cols[i].cpu().numpy().view(np.uint64), blinds[i][None]]), bases=gl, column_class=kinds[done + i]), pts[i])
                checked += 1
                e1 = None
        if e1 is None:  # a hook ran on the host meanwhile (events measure wall time between their records: the idle gap is not the transform's)
            commit_ms = e0.elapsed_time(e1_commit)
            e1 = ev()
        else:
            commit_ms = None
        coeff = dom.lagrange_to_coeff(cols)
        e2 = ev()
        ext = dom.coeff_to_extended_blocks(coeff, D, out=ext_buf) if blocks else dom.coeff_to_extended(coeff, out=ext_buf)
        e3 = ev()
        # the evaluations at the challenge x that precede the multiopen argument (eval_polynomial per queried column;
        # the real prover does them after x is squeezed, with the coefficient forms kept resident: same work)
        evals = api.poly_eval_batch_dev(field, coeff, n, b, x_eval, stream=torch.cuda.current_stream().cuda_stream)
        e4 = ev()
        torch.cuda.synchronize()
        times["evals"] += e3.elapsed_time(e4)
        counts["evals"] += b
        times["commit_lagrange"] += commit_ms if commit_ms is not None else e0.elapsed_time(e1)
        times["lagrange_to_coeff"] += e1.elapsed_time(e2)
        times["coeff_to_extended"] += e2.elapsed_time(e3)
        counts["commit_lagrange"] += b
        counts["lagrange_to_coeff"] += b
        counts["coeff_to_extended"] += b
        if hook is not None and done == 0:
            torch.cuda.synchronize()
            for i in range(min(b, 3)):
                hook("commit_lagrange", dict(scalars=np.concatenate([cols_h[i], blinds[i][None]]), bases=gl), pts[i])
                hook("lagrange_to_coeff", dict(a=cols_h[i], domain=(field, QUOTIENT_J, k)), coeff[i].cpu().numpy().view(np.uint64))
                hook("coeff_to_extended_blocks" if blocks else "coeff_to_extended", dict(a=coeff[i].cpu().numpy().view(np.uint64), domain=(field, QUOTIENT_J, k), n_blocks=D),
                     ext[i].cpu().numpy().view(np.uint64))
                hook("evals", dict(a=coeff[i].cpu().numpy().view(np.uint64), x=x_eval, field=field), evals[i])
                checked += 4
        if done + b >= lag_total:
            ext_keep = ext  # the last batch of extended cosets stays resident for the h(X) step below
        assert coeff.data_ptr() == cols.data_ptr()  # in place: the rows of coeff_all now hold the coefficient forms
        del coeff, cols
        done += b

    # the same per-column phase with the transforms on a second context / stream / host thread (reported beside the step-by-step sum)
    loop_ms = None
    if overlap:
        keep_ext = ext_keep.clone()
        batches, bls, d0 = [], [], 0
        while d0 < lag_total:
            b = min(batch, lag_total - d0)
            batches.append(make_columns(d0, b)[1])
            bls.append(synth.field_elements(seed + 0x100000 + d0, b))
            d0 += b
        # each form twice over fresh copies, the faster one counts: the first pass over a new context / stream also pays one-off costs of
        # the HIP runtime (a 37 ms stall inside one launch was measured at a fixed position of the first pass)
        loop_ms = {}
        for name, ovl in (("step_by_step", False), ("two_contexts_overlapped", True)):
            best = None
            for _ in range(2):
                copies = [t.clone() for t in batches]
                ms = _column_loop_two_contexts(params, dom, copies, bls, ext_buf, D, blocks, x_eval, field, n, dev, ovl)
                del copies
                best = ms if best is None else min(best, ms)
            loop_ms[name] = round(best, 3)
        del batches
        ext_keep = keep_ext

    # --- h(X) numerator: gate expressions over the extended cosets, folded with the challenge y.  The real gate set is the
    # reference's circuit definition (src/circuits/tables/exe.rs:147-498 etc.), which cannot be extracted without the Rust
    # toolchain: a synthetic set with the same shape (selector-gated constraints up to degree 6) over the resident columns ---
    nres = ext_keep.shape[0]
    gates = expr.synthetic_gates(n_advice=max(1, nres - 4), n_fixed=min(4, nres), n_gates=N_SYNTH_GATES)
    prog = expr.compile_gates(field, gates, y=0x5EED)
    res = {}
    for i, key in enumerate(sorted(prog.columns)):
        res[key] = ext_keep[i % nres]
    gev = expr.GateEvaluator(prog)
    e0 = ev()
    h_num = gev.eval_blocks(res, k, D) if blocks else gev.eval(res, ek, 1 << (ek - k))
    e1 = ev()
    torch.cuda.synchronize()
    times["h_eval"] += e0.elapsed_time(e1)
    counts["h_eval"] += 1
    if hook is not None:
        if blocks:
            hook("h_eval", dict(gates=gates, resident=res, block_log=k, n_blocks=D, y=0x5EED, field=field), h_num)
        else:
            hook("h_eval", dict(gates=gates, resident=res, log_n=ek, rot_step=1 << (ek - k), y=0x5EED, field=field), h_num)
        checked += 1
    # the same step over the REFERENCE'S gate polynomials (gateset.py: the committed fixtures of all 30 create_gate sites with the
    # circuit's multiplicities: 142 polynomials over 205 advice columns + 3 selectors); every program column is a distinct resident
    # extended column (copies of this batch's), so the evaluator's reads are the real set's reads.  Reported beside the synthetic figure.
    real = None
    from . import gateset
    ref_gates = gateset.reference_gates(gates_dir)  # the fixtures' directory is the caller's to name (tests/golden in this repository)
    if ref_gates is not None:
        rgates, rinfo = ref_gates
        rprog = expr.compile_gates(field, rgates, y=0x5EED)
        rcols = torch.empty((len(rprog.columns), ext_rows, 4), dtype=torch.int64, device=dev)
        for i in range(rcols.shape[0]):
            rcols[i].copy_(ext_keep[i % nres].reshape(ext_rows, 4))
        rres = {key: rcols[i] for i, key in enumerate(sorted(rprog.columns))}
        rev = expr.GateEvaluator(rprog)
        rout = torch.empty((1, ext_rows, 4), dtype=torch.int64, device=dev)
        rev.eval_blocks(rres, k, D, out=rout) if blocks else rev.eval(rres, ek, 1 << (ek - k), out=rout)  # untimed first call
        e0 = ev()
        h_real = rev.eval_blocks(rres, k, D, out=rout) if blocks else rev.eval(rres, ek, 1 << (ek - k), out=rout)
        e1 = ev()
        torch.cuda.synchronize()
        real = {"ms": round(e0.elapsed_time(e1), 3), "gates": rinfo["gates"], "columns": len(rprog.columns), "instructions": len(rprog.insns),
                "degree_histogram": rinfo["degree_histogram"], "by_site": rinfo["by_site"]}
        if hook is not None:
            if blocks:
                hook("h_eval", dict(gates=rgates, resident=rres, block_log=k, n_blocks=D, y=0x5EED, field=field, real=True), h_real)
            else:
                hook("h_eval", dict(gates=rgates, resident=rres, log_n=ek, rot_step=1 << (ek - k), y=0x5EED, field=field, real=True), h_real)
            checked += 1
        del rcols, rres, rout, h_real, rev
    del ext_keep, ext_buf, res, h_num

    # --- coefficient-basis commits: vanishing random poly, h pieces ---
    ncoef = 1 + N_H_PIECES
    cols_h = synth.field_elements(0xABC, ncoef * n).reshape(ncoef, n, 4)
    blinds = synth.field_elements(0xABD, ncoef)
    cols = torch.from_numpy(cols_h.view(np.int64)).to(dev)
    coeff_all[lag_total:].copy_(cols)
    e0 = ev()
    pts = params.commit_batch(cols, blinds)
    e1 = ev()
    torch.cuda.synchronize()
    times["commit"] += e0.elapsed_time(e1)
    counts["commit"] += ncoef
    if hook is not None:
        hook("commit", dict(scalars=np.concatenate([cols_h[0], blinds[0][None]]), bases=g), pts[0])
        checked += 1

    # --- h(X): one extended iNTT ---
    h_h = synth.field_elements(0xEE, ext_rows)
    h = torch.from_numpy(h_h.view(np.int64)).to(dev).reshape(1, ext_rows, 4)
    e0 = ev()
    if blocks:
        hc = dom.blocks_to_quotient(h, divide_by_vanishing=True).reshape(1, D * n, 4)
    else:
        dom.divide_by_vanishing_poly(h)
        hc = dom.extended_to_coeff(h)
    e1 = ev()
    torch.cuda.synchronize()
    times["extended_to_coeff"] += e0.elapsed_time(e1)
    counts["extended_to_coeff"] += 1
    if hook is not None:
        hook("blocks_to_quotient" if blocks else "divide_and_extended_to_coeff", dict(a=h_h, domain=(field, QUOTIENT_J, k), n_blocks=D), hc[0].cpu().numpy().view(np.uint64))
        checked += 1

    # --- poly::multiopen::create_proof over everything the prover opened: every column at x; the rotated advice queries and the
    # product columns also at omega x, the permuted lookup inputs at omega^-1 x, the permutation products (but the last) at
    # omega^last x -- four point sets; then the IPA opening of the folded polynomial.  Polynomials of one point set are stored
    # back to back so that the x1 fold reads them in place ---
    m = poly._MODULUS[field]
    x = int.from_bytes(synth.field_elements(0xE7A, 1)[0].tobytes(), "little") % m
    w = dom.omega
    xw, xwi, xlast = x * w % m, x * pow(w, -1, m) % m, x * pow(w, n - BLINDING_ROWS - 1, m) % m
    c0 = N_INSTANCE                                   # advice starts here
    rot = range(c0, c0 + 40)                          # advice columns queried at the next row as well (program counter, registers, flags)
    lk0 = N_INSTANCE + N_ADVICE                       # permuted lookup columns: input / table pairs
    z0 = lk0 + 2 * N_LOOKUPS                          # lookup products, then permutation products
    queries = []
    for c in range(coeff_all.shape[0]):
        queries.append((x, c))
        if c in rot or c >= z0 and c < lag_total:
            queries.append((xw, c))
        if lk0 <= c < z0 and (c - lk0) % 2 == 0:
            queries.append((xwi, c))
        if z0 + N_LOOKUPS <= c < lag_total - 1:
            queries.append((xlast, c))
    # group the polynomials by point set, contiguously (a permutation of the rows: which column is which does not matter any more)
    commitments, point_sets = multiopen.construct_intermediate_sets(queries)
    order = sorted(range(len(commitments)), key=lambda i: commitments[i][1])
    remap = {commitments[i][0]: pos for pos, i in enumerate(order)}
    queries = [(pt, remap[c]) for pt, c in queries]
    queries.sort(key=lambda q: q[1])  # first appearance in storage order: every set's members are consecutive rows
    polys = {c: coeff_all[c] for c in range(coeff_all.shape[0])}
    blinds_mo = {c: (0x1000 + c) % m for c in polys}
    draws = iter(range(7, 10 ** 9, 13))
    s_h = synth.field_elements(0x5A, n)  # the prover's random s(X), drawn on the host
    ipa_ms = []
    real_ipa = ipa.create_proof_native

    def timed_ipa(*a, **kw):
        t0 = ev()
        r = real_ipa(*a, **kw)
        t1 = ev()
        torch.cuda.synchronize()
        ipa_ms.append(t0.elapsed_time(t1))
        return r

    torch.cuda.synchronize()
    e0 = ev()
    ipa.create_proof_native = timed_ipa
    try:
        multiopen.create_proof(params, lambda: next(draws), _FixedTranscript(m), queries, polys, blinds_mo, s_poly=s_h)
    finally:
        ipa.create_proof_native = real_ipa
    e1 = ev()
    torch.cuda.synchronize()
    times["ipa"] += ipa_ms[0]
    times["multiopen_folds"] += e0.elapsed_time(e1) - ipa_ms[0]
    counts["ipa"] += 1
    counts["multiopen_folds"] += len(point_sets)
    del coeff_all, polys

    # --- keygen_vk / keygen_pk, once per proving key (not part of gpu_ms_total): fixed and sigma columns are committed, brought to
    # coefficient form and extended to the coset; l0, l_blind, l_last are extended (/root/reference/src/test_utils.rs:23-25).
    # Fixed columns are selectors / range tables (small values), sigma columns full-size field elements ---
    keygen_ms = None
    if keygen:
        kg = {"commit_lagrange": 0.0, "lagrange_to_coeff": 0.0, "coeff_to_extended": 0.0}
        kext = torch.empty((min(batch, N_FIXED + N_SIGMA), ext_rows, 4), dtype=torch.int64, device=dev)
        todo = [(N_FIXED, "flag" if columns == "witness" else "full"), (N_SIGMA, "full"), (3, "flag" if columns == "witness" else "full")]
        for count, kind in todo:
            first = 0
            while first < count:
                b = min(batch, count - first)
                can = witness_columns(kind, False, 0x6E9 + first, b, n, word_bits)
                d = torch.from_numpy(can.view(np.int64)).to(dev)
                if kind != "full":
                    api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS["to_mont"], api._devptr(d), None, api._devptr(d), b * n, stream))
                t0 = ev()
                if count != 3:  # l0 / l_blind / l_last are not committed
                    params.commit_lagrange_batch(d, synth.field_elements(0x6EA + first, b))
                t1 = ev()
                cf = dom.lagrange_to_coeff(d)
                t2 = ev()
                if blocks:
                    dom.coeff_to_extended_blocks(cf, D, out=kext)
                else:
                    dom.coeff_to_extended(cf, out=kext)
                t3 = ev()
                torch.cuda.synchronize()
                kg["commit_lagrange"] += t0.elapsed_time(t1)
                kg["lagrange_to_coeff"] += t1.elapsed_time(t2)
                kg["coeff_to_extended"] += t2.elapsed_time(t3)
                first += b
        keygen_ms = {kk: round(v, 3) for kk, v in kg.items()}
        keygen_ms["total"] = round(sum(kg.values()), 3)
        keygen_ms["columns"] = {"fixed": N_FIXED, "sigma": N_SIGMA, "l0_l_blind_l_last": 3}
        del kext

    wall = time.perf_counter() - t_wall
    out = {"word_bits": word_bits, "columns": columns, "extended_domain": f"{D} of {1 << (ek - k)} coset blocks of 2^{k}" if blocks else f"all 2^{ek} points", "schedule": sch, "counts": counts, "gpu_ms": {kk: round(v, 3) for kk, v in times.items()},
           "gpu_ms_total": round(sum(times.values()), 3),
           "column_loop_ms": loop_ms,
           "gpu_ms_total_two_contexts": round(sum(times.values()) - (loop_ms["step_by_step"] - loop_ms["two_contexts_overlapped"]), 3) if loop_ms else None,
           "h_eval_synthetic_gates": N_SYNTH_GATES, "h_eval_real_gates_ms": real["ms"] if real else None, "h_eval_real_gates": real,
           "gpu_ms_total_with_real_gates": round(sum(times.values()) - times["h_eval"] + real["ms"], 3) if real else None,
           "scope": "GPU time of the offloaded arithmetic of ONE create_proof incl. the multiopen folds / divisions; witness generation, the transcript and PCIe are not in it",
           "keygen_gpu_ms": keygen_ms, "fixed_base_tables": bool(precompute), "setup_precompute_ms": round(precompute_ms, 3),
           "wall_s_including_host_input_generation": round(wall, 3), "checked_against_oracle": checked, "clock_warmup": warm}
    if verbose:
        print(json.dumps(out))
    return out


def run_dropin(word_bits: int, level: str = "literal", batch: int = 64, hook=None, device: int = 0, verbose: bool = True, columns: str = "witness",
               max_columns: int | None = None) -> dict:
    """The same schedule with every polynomial in HOST memory, as the reference's Rust `create_proof` keeps them
    (/root/reference/src/test_utils.rs:41-49): north_star's literal integration, priced with PCIe.

    level "literal": the two-function seam.  Per column `Params::commit_lagrange` -> trh_msm (resident bases, host scalars),
        `lagrange_to_coeff` -> trh_best_fft (2^k, both ways over the link), `coeff_to_extended` -> trh_best_fft on the zero-padded
        2^extended_k host vector (64 MiB each way at k = 18); h(X) -> one trh_best_fft of 2^extended_k; the IPA's round MSMs ->
        trh_best_multiexp over the host's folded generators.  One synchronous call at a time, as the Rust loops issue them.
    level "batched-blocks": as "batched", for a host whose h(X) evaluation has adopted the coset-block layout of the extended domain
        (trh_domain_coeff_to_extended_blocks_host / trh_domain_blocks_to_quotient_host): 5 x 2^k values per column come down instead of 2^extended_k.
    level "batched": the Params / EvaluationDomain seam with column batches (trh_commit_batch_host,
        trh_domain_lagrange_to_coeff_host, trh_domain_coeff_to_extended_host: only the 2^k coefficients go up, uploads /
        kernels / downloads of consecutive columns overlap), h(X) through trh_domain_extended_to_coeff_host, the opening
        through the single-call IPA with p(X), s(X) uploaded.
    What stays on the Rust host in both (and is NOT timed here): the pointwise steps of EvaluationDomain in the literal level
    (x n^-1, the zeta shift), the lookup / permutation products, h(X)'s gate evaluation, the transcript.  The values passed on
    between steps are therefore not the prover's (timing does not depend on them); `hook` sees inputs and outputs of each call."""
    assert level in ("literal", "batched", "batched-blocks") and columns in ("random", "witness")
    import torch
    blocks_level = level == "batched-blocks"  # as "batched", with the extended domain in the coset-block layout: 5/8 of the bytes come down
    if blocks_level:
        level = "batched"

    k = 2 + word_bits // 2
    sch = schedule(k)
    n, ek = sch["n"], sch["extended_k"]
    N = 1 << ek
    curve, field = "vesta", "fp"
    api.init(device)
    dev = torch.device("cuda", device)
    dom = poly.EvaluationDomain(field, QUOTIENT_J, k)
    g = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n + 1)
    gl = api.Bases.generate(curve, synth.BASE_S0 + 77, synth.BASE_D + 2, n + 1)
    params = poly.Params.__new__(poly.Params)
    params.curve, params.k, params.n = curve, k, n
    params._g, params._g_lagrange = g, gl
    g_host = g.download()
    params.w = g_host[n:n + 1]
    params.u = api.Bases.generate(curve, 4242, 1, 1).download()
    params.precompute()
    lag_total = sch["intt_n"] if max_columns is None else min(max_columns, sch["intt_n"])
    kinds = [(kind, blinded) for count, kind, blinded in column_classes(word_bits) for _ in range(count)]
    stream = torch.cuda.current_stream().cuda_stream

    def host_columns(first, b, seed=0xC01):
        """b host columns (Montgomery limbs), each its own C-contiguous (n, 4) array -- separate allocations, as `Vec<F>`s are"""
        if columns == "random":
            return [np.ascontiguousarray(synth.field_elements(seed + first + i, n)) for i in range(b)]
        can = np.empty((b, n, 4), dtype=np.uint64)
        i = 0
        while i < b:
            j = i
            while j < b and kinds[first + j] == kinds[first + i]:
                j += 1
            can[i:j] = witness_columns(kinds[first + i][0], kinds[first + i][1], seed + first + i, j - i, n, word_bits)
            i = j
        d = torch.from_numpy(can.view(np.int64)).to(dev)
        api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS["to_mont"], api._devptr(d), None, api._devptr(d), b * n, stream))
        torch.cuda.synchronize()
        h = d.cpu().numpy().view(np.uint64).reshape(b, n, 4)
        return [np.ascontiguousarray(h[i]) for i in range(b)]

    wall = {"commit_lagrange": 0.0, "lagrange_to_coeff": 0.0, "coeff_to_extended": 0.0, "commit": 0.0, "extended_to_coeff": 0.0, "ipa": 0.0}
    counts = {kk: 0 for kk in wall}
    checked = 0
    w_inv, w_ext, w_ext_inv = dom._w["omega_inv"], dom._w["extended_omega"], dom._w["extended_omega_inv"]
    api.io_stats(reset=True)
    t_all = time.perf_counter()
    done = 0
    ext_bufs = None
    while done < lag_total:
        b = min(batch, lag_total - done)
        cols = host_columns(done, b)
        blinds = synth.field_elements(0x100C01 + done, b)
        if level == "literal":
            if ext_bufs is None:
                ext_bufs = [np.zeros((N, 4), dtype=np.uint64) for _ in range(2)]
            for i in range(b):
                sc = np.concatenate([cols[i], blinds[i][None]])  # the Rust side passes poly || blind through a chained iterator; the copy is not timed
                t0 = time.perf_counter()
                pt = gl.msm(sc)
                t1 = time.perf_counter()
                wall["commit_lagrange"] += t1 - t0
                if hook is not None and done == 0 and i < 3:
                    hook("commit_lagrange", dict(scalars=sc, bases=gl), pt)
                    checked += 1
                a_in = cols[i].copy() if hook is not None and done == 0 and i < 3 else None
                t0 = time.perf_counter()
                api.best_fft_inplace(field, cols[i], w_inv, k)
                t1 = time.perf_counter()
                wall["lagrange_to_coeff"] += t1 - t0
                if a_in is not None:
                    hook("best_fft", dict(a=a_in, omega=w_inv, log_n=k, field=field), cols[i])
                    checked += 1
                ext = ext_bufs[i & 1]
                ext[:n] = cols[i]  # zero-padding (and, on the Rust host, the zeta shift) is host work
                ext[n:] = 0
                e_in = ext[:n].copy() if a_in is not None else None
                t0 = time.perf_counter()
                api.best_fft_inplace(field, ext, w_ext, ek)
                t1 = time.perf_counter()
                wall["coeff_to_extended"] += t1 - t0
                if e_in is not None and i == 0:
                    hook("best_fft_padded", dict(a=e_in, omega=w_ext, log_n=ek, field=field), ext)
                    checked += 1
        else:
            if ext_bufs is None or len(ext_bufs) < b:
                ext_bufs = [np.zeros((QUOTIENT_J - 1, n, 4) if blocks_level else (N, 4), dtype=np.uint64) for _ in range(b)]
            lag_in = [c.copy() for c in cols[:3]] if hook is not None and done == 0 else None
            t0 = time.perf_counter()
            pts = params.commit_lagrange_batch_host(cols, blinds)
            t1 = time.perf_counter()
            dom.lagrange_to_coeff_host(cols)
            t2 = time.perf_counter()
            if blocks_level:
                dom.coeff_to_extended_blocks_host(cols, QUOTIENT_J - 1, out=ext_bufs[:b])
            else:
                dom.coeff_to_extended_host(cols, out=ext_bufs[:b])
            t3 = time.perf_counter()
            wall["commit_lagrange"] += t1 - t0
            wall["lagrange_to_coeff"] += t2 - t1
            wall["coeff_to_extended"] += t3 - t2
            if lag_in is not None:
                for i in range(len(lag_in)):
                    hook("commit_lagrange", dict(scalars=np.concatenate([lag_in[i], blinds[i][None]]), bases=gl), pts[i])
                    hook("lagrange_to_coeff", dict(a=lag_in[i], domain=(field, QUOTIENT_J, k)), cols[i])
                    hook("coeff_to_extended_blocks" if blocks_level else "coeff_to_extended", dict(a=cols[i], domain=(field, QUOTIENT_J, k), n_blocks=QUOTIENT_J - 1), ext_bufs[i])
                    checked += 3
        for kk in ("commit_lagrange", "lagrange_to_coeff", "coeff_to_extended"):
            counts[kk] += b
        done += b

    # coefficient-basis commits: the vanishing argument's random polynomial and the h pieces
    ncoef = 1 + N_H_PIECES
    ccols = [np.ascontiguousarray(synth.field_elements(0xABC + i, n)) for i in range(ncoef)]
    cbl = synth.field_elements(0xABD, ncoef)
    t0 = time.perf_counter()
    if level == "literal":
        cpts = [g.msm(np.concatenate([ccols[i], cbl[i][None]])) for i in range(ncoef)]
    else:
        cpts = params.commit_batch_host(ccols, cbl)
    wall["commit"] += time.perf_counter() - t0
    counts["commit"] += ncoef
    if hook is not None:
        hook("commit", dict(scalars=np.concatenate([ccols[0], cbl[0][None]]), bases=g), cpts[0])
        checked += 1

    # h(X): the quotient's numerator comes back from the host's gate evaluation as 2^extended_k values
    h_h = np.ascontiguousarray(synth.field_elements(0xEE, (QUOTIENT_J - 1) * n if blocks_level else N))
    h_in = h_h.copy() if hook is not None else None
    t0 = time.perf_counter()
    if level == "literal":
        api.best_fft_inplace(field, h_h, w_ext_inv, ek)
    elif blocks_level:
        h_h = dom.blocks_to_quotient_host(h_h, divide_by_vanishing=True)
    else:
        dom.extended_to_coeff_host(h_h, divide_by_vanishing_first=True)
    wall["extended_to_coeff"] += time.perf_counter() - t0
    counts["extended_to_coeff"] += 1
    if hook is not None:
        if level == "literal":
            hook("best_fft", dict(a=h_in, omega=w_ext_inv, log_n=ek, field=field), h_h)
        elif blocks_level:
            hook("blocks_to_quotient", dict(a=h_in, domain=(field, QUOTIENT_J, k), n_blocks=QUOTIENT_J - 1), h_h)
        else:
            hook("divide_and_extended_to_coeff", dict(a=h_in, domain=(field, QUOTIENT_J, k)), h_h[: n * (QUOTIENT_J - 1)])
        checked += 1

    # the opening
    p_h = np.ascontiguousarray(synth.field_elements(0x9A, n))
    s_h = np.ascontiguousarray(synth.field_elements(0x5A, n))
    m = poly._MODULUS[field]
    t0 = time.perf_counter()
    if level == "literal":
        # commitment::create_proof on the host: S = commit(s), then per round two best_multiexp over the halves of p' and of the
        # host's folded generators G' (bases AND scalars cross the link), the (u, w) two-term MSMs, the folds on the host
        g.msm(np.concatenate([s_h, cbl[0][None]]))
        gp = np.ascontiguousarray(g_host[:n])
        for j in range(k):
            half = 1 << (k - j - 1)
            lj = api.best_multiexp(curve, p_h[half:2 * half], gp[:half])
            rj = api.best_multiexp(curve, p_h[:half], gp[half:2 * half])
            if hook is not None and j in (0, k - 1):
                th = time.perf_counter()
                hook("best_multiexp", dict(scalars=p_h[half:2 * half], bases=gp[:half], curve=curve), lj)
                hook("best_multiexp", dict(scalars=p_h[:half], bases=gp[half:2 * half], curve=curve), rj)
                checked += 2
                t0 += time.perf_counter() - th  # the checker's time is not the opening's
    else:
        draws = iter(range(7, 10 ** 9, 13))
        p_dev = torch.from_numpy(p_h.view(np.int64)).to(dev)
        ipa.create_proof_native(params, lambda: next(draws), _FixedTranscript(m), p_dev, 0x1234, int.from_bytes(synth.field_elements(0xE7A, 1)[0].tobytes(), "little") % m, s_h, 0x77)
    wall["ipa"] += time.perf_counter() - t0
    counts["ipa"] += 1

    total_s = time.perf_counter() - t_all
    io = api.io_stats()
    in_calls = sum(wall.values())
    out = {"mode": "dropin-" + level + ("-blocks" if blocks_level else ""), "word_bits": word_bits, "columns": columns, "schedule": sch, "counts": counts,
           "wall_ms_incl_pcie": {kk: round(v * 1e3, 3) for kk, v in wall.items()}, "wall_ms_incl_pcie_total": round(in_calls * 1e3, 3),
           "pcie": {"h2d_GB": round(io["h2d_bytes"] / 1e9, 3), "d2h_GB": round(io["d2h_bytes"] / 1e9, 3),
                    # of h2d_GB: pinned slots that were zero throughout (the padding of the zero-padded 2^extended_k vectors) and were cleared on the
                    # device instead of crossing the link
                    "h2d_zero_elided_GB": round(io.get("h2d_zero_bytes", 0.0) / 1e9, 3),
                    "h2d_GBps_in_copies": round(io["h2d_bytes"] / max(io["h2d_seconds"], 1e-9) / 1e9, 2),
                    "d2h_GBps_in_copies": round(io["d2h_bytes"] / max(io["d2h_seconds"], 1e-9) / 1e9, 2),
                    "GBps_over_call_time": round((io["h2d_bytes"] + io["d2h_bytes"]) / max(in_calls, 1e-9) / 1e9, 2),
                    "link_peak_GBps_per_direction": 57.0, "link_peak_source": "pinned hipMemcpyAsync on the box, profiles/pcie_probe_r03.txt (PCIe Gen5 x16: 63 GB/s spec)"},
           "scope": "wall time inside libtrh's host-pointer calls for ONE create_proof (uploads + kernels + downloads); the Rust host's own work between the calls is not in it",
           "columns_replayed": lag_total, "wall_s_including_host_input_generation": round(total_s, 3), "checked_against_oracle": checked}
    if verbose:
        print(json.dumps(out))
    return out


def run_sharded(word_bits: int, devices, batch: int = 64, columns: str = "witness", verbose: bool = True, max_columns: int | None = None) -> dict:
    """The per-column phase of create_proof (commit_lagrange, lagrange_to_coeff, coeff_to_extended as coset blocks, the evaluations at x)
    COLUMN-SHARDED over a list of devices, in one process -- how a single Rust prover process (the reference proves sequentially in one
    process, /root/reference/src/test_utils.rs:37-54) drives the GPUs of a node: per device one host thread, one libtrh context
    (trh_ctx_create), one Params copy (bases + fixed-base tables resident on that device) and one EvaluationDomain; thread g takes the
    contiguous columns sharded.shard_range(497, g, G), no data-path collective (the columns of a proof are independent;
    96 bytes per column come back for the transcript).  A device may be listed more than once (`--devices 0,0` on a one-GPU box: two
    contexts, two threads, two streams on the same chip).  The commitments, gathered in column order, must equal the single-context
    run's bit for bit; the result carries per-device times and the single-context time of the same phase."""
    import threading

    import torch

    from . import sharded
    assert columns in ("random", "witness")
    devices = [int(d) for d in devices]
    G = len(devices)
    k = 2 + word_bits // 2
    sch = schedule(k)
    n = sch["n"]
    D = QUOTIENT_J - 1
    curve, field = "vesta", "fp"
    api.init(devices[0])
    lag_total = sch["intt_n"] if max_columns is None else min(max_columns, sch["intt_n"])
    kinds = [(kind, blinded) for count, kind, blinded in column_classes(word_bits) for _ in range(count)]
    x_eval = synth.field_elements(0xE7A, 1)[0]
    seed = 0xC01

    class Shard:
        def __init__(self, g, dev_index, lo, hi):
            self.g, self.dev_index, self.lo, self.hi = g, dev_index, lo, hi
            self.dev = torch.device("cuda", dev_index)
            self.ctx = api.Context(dev_index)
            self.ms = {"commit_lagrange": 0.0, "lagrange_to_coeff": 0.0, "coeff_to_extended": 0.0, "evals": 0.0}
            self.points, self.evals, self.err = None, None, None

        def setup(self):
            """bases, tables, domain and this shard's columns, resident on its device (untimed: once per proving key / witness)"""
            self.ctx.bind()
            torch.cuda.set_device(self.dev)
            self.stream = torch.cuda.Stream(device=self.dev)
            params = poly.Params.__new__(poly.Params)
            params.curve, params.k, params.n = curve, k, n
            params._g = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n + 1)
            params._g_lagrange = api.Bases.generate(curve, synth.BASE_S0 + 77, synth.BASE_D + 2, n + 1)
            params.w = params._g.download(n, 1)
            params.u = None
            for b in (params._g, params._g_lagrange):
                b.precompute(0)
            self.params = params
            self.dom = poly.EvaluationDomain(field, QUOTIENT_J, k)
            cnt = self.hi - self.lo
            self.cols = torch.empty((cnt, n, 4), dtype=torch.int64, device=self.dev)
            st = torch.cuda.current_stream().cuda_stream
            i = 0
            while i < cnt:
                if columns == "random":
                    j = min(cnt, i + 16)
                    self.cols[i:j].copy_(torch.from_numpy(synth.field_elements(seed + self.lo + i, (j - i) * n).reshape(j - i, n, 4).view(np.int64)))
                else:
                    j = i
                    while j < cnt and j - i < 32 and kinds[self.lo + j] == kinds[self.lo + i]:
                        j += 1
                    can = witness_columns(kinds[self.lo + i][0], kinds[self.lo + i][1], seed + self.lo + i, j - i, n, word_bits)
                    d = torch.from_numpy(can.view(np.int64)).to(self.dev)
                    api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS["to_mont"], api._devptr(d), None, api._devptr(d), (j - i) * n, st))
                    torch.cuda.synchronize(self.dev)
                    self.cols[i:j].copy_(d)
                    del d
                i = j
            self.blinds = synth.field_elements(seed + 0x100000 + self.lo, max(cnt, 1))[:cnt]
            self.ext = torch.empty((min(batch, max(cnt, 1)), D * n, 4), dtype=torch.int64, device=self.dev)
            torch.cuda.synchronize(self.dev)

        def phase(self, cols, timed=True):
            """this shard's columns through the per-column steps, batch by batch (in place: the rows end as coefficient forms)"""
            cnt = cols.shape[0]
            pts, evs = np.zeros((cnt, 12), dtype=np.uint64), np.zeros((cnt, 4), dtype=np.uint64)
            with torch.cuda.stream(self.stream):
                def ev():
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    return e
                done = 0
                while done < cnt:
                    b = min(batch, cnt - done)
                    c = cols[done:done + b]
                    e0 = ev()
                    pts[done:done + b] = self.params.commit_lagrange_batch(c, self.blinds[done:done + b])
                    e1 = ev()
                    coeff = self.dom.lagrange_to_coeff(c)
                    e2 = ev()
                    self.dom.coeff_to_extended_blocks(coeff, D, out=self.ext)
                    e3 = ev()
                    evs[done:done + b] = api.poly_eval_batch_dev(field, coeff, n, b, x_eval, stream=self.stream.cuda_stream)
                    e4 = ev()
                    self.stream.synchronize()
                    if timed:
                        for key, (a, z) in (("commit_lagrange", (e0, e1)), ("lagrange_to_coeff", (e1, e2)), ("coeff_to_extended", (e2, e3)), ("evals", (e3, e4))):
                            self.ms[key] += a.elapsed_time(z)
                    done += b
            return pts, evs

        def run(self, barrier):
            try:
                self.ctx.bind()
                torch.cuda.set_device(self.dev)
                warm = self.cols[: min(batch, self.cols.shape[0])].clone()
                self.phase(warm, timed=False)   # tables of the transforms, scratch of the MSM (sized by the batch): once per context
                del warm
                work = self.cols.clone()
                torch.cuda.synchronize(self.dev)
                barrier.wait()
                t0 = time.perf_counter()
                self.points, self.evals = self.phase(work)
                torch.cuda.synchronize(self.dev)
                self.wall_ms = (time.perf_counter() - t0) * 1e3
            except Exception as exc:  # surfaced by the caller
                self.err = exc
                try:
                    barrier.abort()
                except Exception:
                    pass
            finally:
                api.Context.unbind()

    shards = []
    for g, dv in enumerate(devices):
        lo, hi = sharded.shard_range(lag_total, g, G)
        shards.append(Shard(g, dv, lo, hi))
    try:
        for sh in shards:   # one after the other: the host-side column generation is the slow part and is not what is measured
            sh.setup()
            api.Context.unbind()
        # the single-context reference: shard 0's context runs ALL columns step by step (the columns of the other shards are copied over)
        ref = shards[0]
        ref.ctx.bind()
        torch.cuda.set_device(ref.dev)
        all_cols = torch.cat([sh.cols.to(ref.dev) for sh in shards]) if G > 1 else ref.cols.clone()
        keep_blinds, keep_ms = ref.blinds, ref.ms
        ref.blinds = np.concatenate([sh.blinds for sh in shards])
        ref.phase(all_cols[: min(batch, lag_total)].clone(), timed=False)
        ref.ms = {kk: 0.0 for kk in keep_ms}
        torch.cuda.synchronize(ref.dev)
        t0 = time.perf_counter()
        ref_pts, ref_evals = ref.phase(all_cols)
        torch.cuda.synchronize(ref.dev)
        single_wall = (time.perf_counter() - t0) * 1e3
        single_ms = dict(ref.ms)
        ref.blinds, ref.ms = keep_blinds, {kk: 0.0 for kk in keep_ms}
        del all_cols
        api.Context.unbind()
        # the sharded run: one thread per device, started together
        barrier = threading.Barrier(G + 1)
        ths = [threading.Thread(target=sh.run, args=(barrier,)) for sh in shards]
        for th in ths:
            th.start()
        try:
            barrier.wait()
        except threading.BrokenBarrierError:
            pass
        t0 = time.perf_counter()
        for th in ths:
            th.join()
        wall = (time.perf_counter() - t0) * 1e3
        for sh in shards:
            if sh.err is not None:
                raise sh.err
        pts = np.concatenate([sh.points for sh in shards])
        evs = np.concatenate([sh.evals for sh in shards])
        identical = bool((pts == ref_pts).all() and (evs == ref_evals).all())
    finally:
        for sh in shards:
            for b in (getattr(getattr(sh, "params", None), "_g", None), getattr(getattr(sh, "params", None), "_g_lagrange", None)):
                if b is not None:
                    b.destroy()
            sh.cols = sh.ext = None
            sh.ctx.destroy()
    out = {"mode": "column-sharded", "word_bits": word_bits, "columns": columns, "devices": devices, "columns_replayed": lag_total,
           "per_device": [{"device": sh.dev_index, "columns": [sh.lo, sh.hi], "wall_ms": round(sh.wall_ms, 3), "gpu_ms": {kk: round(v, 3) for kk, v in sh.ms.items()}} for sh in shards],
           "wall_ms": round(wall, 3), "single_context": {"wall_ms": round(single_wall, 3), "gpu_ms": {kk: round(v, 3) for kk, v in single_ms.items()}},
           "commitments_identical_to_single_context": identical,
           "scope": "per-column phase of ONE create_proof (commit_lagrange, lagrange_to_coeff, coeff_to_extended as 5 coset blocks, evaluations), columns resident; "
                    "one host thread + libtrh context + Params copy per listed device, contiguous column ranges, no collective"}
    if verbose:
        print(json.dumps(out))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--word-bits", type=int, default=32)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-precompute", action="store_true", help="commit over the plain per-window path (no fixed-base tables)")
    ap.add_argument("--columns", choices=("random", "witness"), default="random", help="uniformly random columns, or the value classes of the reference's witness")
    ap.add_argument("--no-keygen", action="store_true")
    ap.add_argument("--mode", choices=("resident", "dropin", "dropin-batched", "dropin-batched-blocks"), default="resident",
                    help="resident: polynomials live on the device (the restructured prover); dropin: every polynomial in host memory, one "
                         "trh_msm / trh_best_fft call at a time (north_star's literal integration); dropin-batched: host memory, batched host-pointer entries")
    ap.add_argument("--extended", choices=("blocks", "full"), default="blocks", help="resident mode: the extended domain as the 5 coset blocks the quotient needs, or all 2^extended_k points")
    ap.add_argument("--overlap", action="store_true", help="resident mode: also time the per-column phase with the transforms on a second context / stream / host thread")
    ap.add_argument("--max-columns", type=int, default=None, help="drop-in modes / --devices: replay only the first N Lagrange columns")
    repo_golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")  # the command line's default, not the package's
    ap.add_argument("--gates-dir", default=repo_golden if os.path.isdir(repo_golden) else None,
                    help="directory with exe_tempvar_gates.json / chip_gates.json: h(X) is then also timed on the reference's gate polynomials")
    ap.add_argument("--devices", default=None, help="comma-separated device list (a device may repeat): the per-column phase column-sharded over one context + thread per entry")
    a = ap.parse_args()
    if a.devices is not None:
        r = run_sharded(a.word_bits, [int(v) for v in a.devices.split(",")], a.batch, columns=a.columns, max_columns=a.max_columns)
        sys.exit(0 if r["commitments_identical_to_single_context"] else 1)
    if a.mode != "resident":
        run_dropin(a.word_bits, {"dropin": "literal", "dropin-batched": "batched", "dropin-batched-blocks": "batched-blocks"}[a.mode], a.batch, columns=a.columns, max_columns=a.max_columns)
        return
    run(a.word_bits, a.batch, precompute=not a.no_precompute, columns=a.columns, keygen=not a.no_keygen, extended=a.extended, overlap=a.overlap, gates_dir=a.gates_dir)


if __name__ == "__main__":
    main()
