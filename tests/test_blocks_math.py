"""CPU tests (no GPU): the identities the coset-block form of the extended domain (csrc/domain.hip, DESIGN K4b) rests on, stated with the
oracle's own EvaluationDomain / best_fft only, and the reference gate set the replay times h(X) on (tiny-ram-halo2_amd/gateset.py)."""
import random

import pytest

import pasta as o
from tiny_ram_halo2_amd import expr, gateset


@pytest.mark.parametrize("field,k,j", [("fp", 3, 6), ("fq", 4, 4), ("fp", 5, 3), ("fq", 3, 8)])
def test_extended_domain_is_cosets_of_the_small_subgroup(field, k, j):
    """(1) entry q 2^(ek-k) + r of coeff_to_extended(a) is the size-n transform of a_i (zeta w_ext^r)^i at q;
    (2) h(X) of degree < (j-1) n is recovered from j - 1 blocks: the inverse coset transform of block r is sum_i c_r^i h_i(X) with
        c_r = (zeta w_ext^r)^n, and the Vandermonde system in c_r has a unique solution -- what trh_domain_blocks_to_quotient computes;
    (3) on block r the vanishing polynomial X^n - 1 is the constant c_r - 1 (EvaluationDomain::t_evaluations has period 2^(ek-k))."""
    f = o.FIELDS[field]
    m = f.m
    dom = o.EvaluationDomain(f, j, k)
    n, ek = 1 << k, dom.extended_k
    step, D = 1 << (ek - k), j - 1
    rng = random.Random(k * 100 + j)
    a = [rng.randrange(m) for _ in range(n)]
    ext = dom.coeff_to_extended(a)
    gens = [dom.g_coset * pow(dom.extended_omega, r, m) % m for r in range(step)]
    for r in range(step):
        scaled = [a[i] * pow(gens[r], i, m) % m for i in range(n)]
        assert o.best_fft(f, scaled, dom.omega, k) == ext[r::step], r
        assert dom.t_evaluations[r] == f.inv((pow(gens[r], n, m) - 1) % m)
    # (2)
    h = [rng.randrange(m) for _ in range(D * n)]
    padded = dom._distribute_powers_zeta(h + [0] * (dom.extended_len() - D * n), True)
    h_ext = o.best_fft(f, padded, dom.extended_omega, ek)
    c = [pow(g, n, m) for g in gens[:D]]
    P = []
    for r in range(D):
        vals = h_ext[r::step]
        coeffs = [v * dom.ifft_divisor % m for v in o.best_fft(f, vals, dom.omega_inv, k)]
        P.append([coeffs[i] * pow(f.inv(gens[r]), i, m) % m for i in range(n)])     # back from the coset: h mod (X^n - c_r)
        assert P[-1] == [sum(pow(c[r], i, m) * h[i * n + e] for i in range(D)) % m for e in range(n)]
    # Gauss-Jordan inverse of V[r][i] = c_r^i over the field
    V = [[pow(c[r], i, m) for i in range(D)] for r in range(D)]
    inv = [[int(r == i) for i in range(D)] for r in range(D)]
    for col in range(D):
        piv = next(r for r in range(col, D) if V[r][col])
        V[col], V[piv], inv[col], inv[piv] = V[piv], V[col], inv[piv], inv[col]
        s = f.inv(V[col][col])
        V[col] = [v * s % m for v in V[col]]
        inv[col] = [v * s % m for v in inv[col]]
        for r in range(D):
            if r != col and V[r][col]:
                t = V[r][col]
                V[r] = [(x - t * y) % m for x, y in zip(V[r], V[col])]
                inv[r] = [(x - t * y) % m for x, y in zip(inv[r], inv[col])]
    got = [sum(inv[i][r] * P[r][e] for r in range(D)) % m for i in range(D) for e in range(n)]
    assert got == h
    # and the oracle's own chain on the numerator h (X^n - 1) gives the same polynomial
    num = [v * f.inv(dom.t_evaluations[i % step]) % m for i, v in enumerate(h_ext)]
    assert dom.extended_to_coeff(dom.divide_by_vanishing_poly(num)) == h


def test_reference_gate_set():
    """the 30 create_gate sites of the reference with the circuit's multiplicities (fixtures tests/golden/*.json + the three small identities
    of gateset.py): counts, degrees (the sprod gate is what makes the quotient degree 5), and that the set lowers to one stack program"""
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    assert gateset.reference_gates() is None or os.environ.get(gateset.GATES_DIR_ENV)  # the package reaches into no directory of its own accord
    got = gateset.reference_gates(golden)
    assert got is not None
    gates, info = got
    assert info["gates"] == 142 and info["max_degree"] == 6
    assert info["by_site"]["exe.rs temp-var / trace gates"] == 90 and info["by_site"]["even_bits"] == gateset.N_EVEN_BITS_CONFIGS == 14
    assert info["by_site"]["signed"] == 2 * gateset.N_SIGNED_CONFIGS and info["by_site"]["sprod"] == 1 and info["by_site"]["logic"] == 5
    assert sum(info["degree_histogram"].values()) == 142 and info["degree_histogram"][6] == 1
    prog = expr.compile_gates("fp", gates, y=7)
    assert prog.max_degree == 6 and len(prog.columns) == info["advice"] + info["selectors"]
    # every gate vanishes on the all-zero assignment except none: a gate polynomial has no constant term iff its selector product does
    f = o.FIELDS["fp"]

    def ev(e, vals):
        if isinstance(e, expr.Constant):
            return e.value % f.m
        if isinstance(e, expr._Query):
            return vals(e)
        if isinstance(e, expr.Negated):
            return -ev(e.e, vals) % f.m
        if isinstance(e, expr.Sum):
            return (ev(e.a, vals) + ev(e.b, vals)) % f.m
        if isinstance(e, expr.Product):
            return ev(e.a, vals) * ev(e.b, vals) % f.m
        return ev(e.e, vals) * e.value % f.m

    assert all(ev(g, lambda q: 0) == 0 for g in gates)           # selectors off: nothing is constrained
    assert any(ev(g, lambda q: 1) != 0 for g in gates)           # and the set is not identically zero
    assert gateset.reference_gates("/nonexistent") is None
