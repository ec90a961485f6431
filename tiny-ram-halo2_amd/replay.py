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

`run(..., hook=f)` calls f(kind, inputs, outputs) on the first items of every primitive kind so that a test
can compare them with the oracle (tests/test_gpu_replay.py); the module itself never touches the oracle.
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np

from . import api, expr, ipa, permutation, poly, synth

# Appendix B counts for TinyRamCircuit<WB, 8>
N_INSTANCE, N_ADVICE, N_LOOKUPS, N_PERM_PRODUCTS, N_H_PIECES = 94, 263, 31, 47, 5
N_SYNTH_GATES = 300  # gate polynomials of the synthetic h(X) step (the real count is a property of the circuit definition)
QUOTIENT_J = 6  # cs.degree() = 6 => EvaluationDomain::new(6, k): quotient_poly_degree 5, extended_k = k + 3


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


def run(word_bits: int, batch: int = 64, hook=None, device: int = 0, verbose: bool = True, precompute: bool = True) -> dict:
    import torch

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
    torch.cuda.synchronize()
    precompute_ms = (time.perf_counter() - t_pre) * 1e3

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    times = {"lookup_permute": 0.0, "product_columns": 0.0, "commit_lagrange": 0.0, "lagrange_to_coeff": 0.0, "coeff_to_extended": 0.0, "evals": 0.0, "h_eval": 0.0, "commit": 0.0,
             "extended_to_coeff": 0.0, "ipa": 0.0}
    counts = {kk: 0 for kk in times}
    checked = 0
    torch.cuda.synchronize()
    t_wall = time.perf_counter()

    # --- lookup argument: permuted input / table columns of the 31 lookups (permute_expression_pair) ---
    distinct = torch.from_numpy(synth.field_elements(0x7AB1E, min(n, 1 << 16)).view(np.int64)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x100C)
    for li in range(N_LOOKUPS):
        table = distinct[torch.arange(n, device=dev) % distinct.shape[0]].contiguous()
        inp = distinct[torch.randint(0, distinct.shape[0], (n,), device=dev, generator=gen)].contiguous()
        e0 = ev()
        pa, ps = permutation.lookup_permute(field, inp, table)
        e1 = ev()
        torch.cuda.synchronize()
        times["lookup_permute"] += e0.elapsed_time(e1)
        counts["lookup_permute"] += 1
        if hook is not None and li == 0:
            hook("lookup_permute", dict(input=inp.cpu().numpy().view(np.uint64), table=table.cpu().numpy().view(np.uint64), field=field),
                 (pa.cpu().numpy().view(np.uint64), ps.cpu().numpy().view(np.uint64)))
            checked += 1
    del distinct

    # --- grand products: the 47 permutation product columns (4 columns each) and the 31 lookup products, all at once ---
    m_ = poly._MODULUS[field]
    beta, gamma = 0xBE7A % m_, 0x6A33A % m_
    wit = [torch.from_numpy(synth.field_elements(0x9E0 + j, n).view(np.int64)).to(dev) for j in range(8)]
    om = torch.empty((n, 4), dtype=torch.int64, device=dev)
    api.powers_dev(field, om, n, expr._limbs(field, permutation.omega(field, k)))
    pcs = [permutation.ProductColumn(field, k, 4, first_column=4 * c) for c in range(N_PERM_PRODUCTS)]
    evs = [pc.evaluator(beta, gamma) for pc in pcs]
    sets = [pc.columns(wit[:4], wit[4:], om) for pc in pcs]
    lk = permutation.lookup_product(field, k, beta, gamma)
    evs += [lk.ev] * N_LOOKUPS
    sets += [{("advice", i): wit[(i + li) % 8] for i in range(4)} for li in range(N_LOOKUPS)]
    e0 = ev()
    zs = permutation.grand_products_batch(field, k, evs, sets)
    e1 = ev()
    torch.cuda.synchronize()
    times["product_columns"] += e0.elapsed_time(e1)
    counts["product_columns"] += len(evs)
    if hook is not None:
        hook("product_column", dict(values=[w.cpu().numpy().view(np.uint64) for w in wit[:4]], sigmas=[w.cpu().numpy().view(np.uint64) for w in wit[4:]],
                                    beta=beta, gamma=gamma, first_column=4, field=field, k=k), zs[1].cpu().numpy().view(np.uint64))
        checked += 1
    del zs, evs, sets, pcs, wit, om

    x_eval = synth.field_elements(0xE7A, 1)[0]
    ext_buf = torch.empty((min(batch, sch["intt_n"]), 1 << ek, 4), dtype=torch.int64, device=dev)  # the batch's extended cosets, reused
    # --- Lagrange-basis columns: instance, advice, lookup permuted x2 + z, permutation z ---
    lag_total = sch["intt_n"]
    done = 0
    seed = 0xC01
    while done < lag_total:
        b = min(batch, lag_total - done)
        cols_h = synth.field_elements(seed + done, b * n).reshape(b, n, 4)
        blinds = synth.field_elements(seed + 0x100000 + done, b)
        cols = torch.from_numpy(cols_h.view(np.int64)).to(dev)
        e0 = ev()
        pts = params.commit_lagrange_batch(cols, blinds)
        e1 = ev()
        coeff = dom.lagrange_to_coeff(cols)
        e2 = ev()
        ext = dom.coeff_to_extended(coeff, out=ext_buf)
        e3 = ev()
        # the evaluations at the challenge x that precede the multiopen argument (eval_polynomial per queried column;
        # the real prover does them after x is squeezed, with the coefficient forms kept resident: same work)
        evals = api.poly_eval_batch_dev(field, coeff, n, b, x_eval, stream=torch.cuda.current_stream().cuda_stream)
        e4 = ev()
        torch.cuda.synchronize()
        times["evals"] += e3.elapsed_time(e4)
        counts["evals"] += b
        times["commit_lagrange"] += e0.elapsed_time(e1)
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
                hook("coeff_to_extended", dict(a=coeff[i].cpu().numpy().view(np.uint64), domain=(field, QUOTIENT_J, k)), ext[i].cpu().numpy().view(np.uint64))
                hook("evals", dict(a=coeff[i].cpu().numpy().view(np.uint64), x=x_eval, field=field), evals[i])
                checked += 4
        if done + b >= lag_total:
            ext_keep = ext  # the last batch of extended cosets stays resident for the h(X) step below
        del coeff, cols
        done += b

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
    h_num = gev.eval(res, ek, 1 << (ek - k))
    e1 = ev()
    torch.cuda.synchronize()
    times["h_eval"] += e0.elapsed_time(e1)
    counts["h_eval"] += 1
    if hook is not None:
        hook("h_eval", dict(gates=gates, resident=res, log_n=ek, rot_step=1 << (ek - k), y=0x5EED, field=field), h_num)
        checked += 1
    del ext_keep, ext_buf, res, h_num

    # --- coefficient-basis commits: vanishing random poly, h pieces ---
    ncoef = 1 + N_H_PIECES
    cols_h = synth.field_elements(0xABC, ncoef * n).reshape(ncoef, n, 4)
    blinds = synth.field_elements(0xABD, ncoef)
    cols = torch.from_numpy(cols_h.view(np.int64)).to(dev)
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
    h_h = synth.field_elements(0xEE, 1 << ek)
    h = torch.from_numpy(h_h.view(np.int64)).to(dev).reshape(1, 1 << ek, 4)
    e0 = ev()
    dom.divide_by_vanishing_poly(h)
    hc = dom.extended_to_coeff(h)
    e1 = ev()
    torch.cuda.synchronize()
    times["extended_to_coeff"] += e0.elapsed_time(e1)
    counts["extended_to_coeff"] += 1
    if hook is not None:
        hook("divide_and_extended_to_coeff", dict(a=h_h, domain=(field, QUOTIENT_J, k)), hc[0].cpu().numpy().view(np.uint64))
        checked += 1

    # --- IPA opening of the final polynomial ---
    m = poly._MODULUS[field]
    p_h = synth.field_elements(0x1FA, n)
    p_dev = torch.from_numpy(p_h.view(np.int64)).to(dev)
    draws = iter(range(7, 10 ** 9, 13))
    s_h = synth.field_elements(0x5A, n)  # the prover's random s(X), drawn on the host
    torch.cuda.synchronize()
    e0 = ev()
    ipa.create_proof_native(params, lambda: next(draws), _FixedTranscript(m), p_dev, 0x1234, 0x77777, s_h, 0x99)
    e1 = ev()
    torch.cuda.synchronize()
    times["ipa"] += e0.elapsed_time(e1)
    counts["ipa"] += 1

    wall = time.perf_counter() - t_wall
    out = {"word_bits": word_bits, "schedule": sch, "counts": counts, "gpu_ms": {kk: round(v, 3) for kk, v in times.items()},
           "gpu_ms_total": round(sum(times.values()), 3), "fixed_base_tables": bool(precompute), "setup_precompute_ms": round(precompute_ms, 3), "wall_s_including_host_input_generation": round(wall, 3),
           "checked_against_oracle": checked}
    if verbose:
        print(json.dumps(out))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--word-bits", type=int, default=32)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-precompute", action="store_true", help="commit over the plain per-window path (no fixed-base tables)")
    a = ap.parse_args()
    run(a.word_bits, a.batch, precompute=not a.no_precompute)


if __name__ == "__main__":
    main()
