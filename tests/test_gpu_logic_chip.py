"""The h(X) / lookup pipeline on a constraint system transcribed from the reference's own chips, with a satisfying witness.

Constraint system (the polynomial identities only, restated in the Expression mirror; WORD_BITS = 8 to keep the witness small):
  * even-bits decomposition, /root/reference/src/circuits/tables/even_bits.rs:143-170 -- for each decomposed word:
      gate    s_table (s_and + s_xor + s_or) (even + 2 odd - word)
      lookups s_table (s_and + s_xor + s_or) even  in  EvenBitsTable,   ... odd  in  EvenBitsTable
  * logic chip, /root/reference/src/circuits/logic.rs:125-185 -- a, b, even_sum, odd_sum decomposed as above and
      l_add   s (a.even + b.even - even_sum.word),  s (a.odd + b.odd - odd_sum.word)
      and     s_table s_and (even_sum.odd + 2 odd_sum.odd - res)
      xor     s_table s_xor (even_sum.even + 2 odd_sum.even - res)
      or      s_table s_or  (even_sum.even + 2 odd_sum.even + even_sum.odd + 2 odd_sum.odd - res)
The reference pins these through MockProver / prove-verify round trips (logic.rs:509-512 all_16_bit_words_test and the negative
tests of test_utils.rs:73-119); here the same statements are made on the device path:
  (1) the folded gates vanish on every row of the witness, and a corrupted cell makes exactly its row non-zero;
  (2) Lagrange -> coeff -> extended coset -> gate evaluation -> divide_by_vanishing_poly -> extended_to_coeff gives an h(X) of
      degree < 2n with h(x) (x^n - 1) == folded gates at a random x;
  (3) every lookup's permuted columns satisfy the argument's constraints, equal the oracle's, and the product column closes;
  (4) a witness whose even part is not an even-bits value is rejected by the lookup (the reference's MockProver failure)."""
import random

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import pasta as o
from tiny_ram_halo2_amd import api, expr, permutation, poly

WORD_BITS = 8
FIELD = "fp"
K = 9


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init(0)
    yield


def spread_even(v):  # the bits of v moved to the even positions
    r = 0
    for b in range(WORD_BITS // 2):
        r |= ((v >> b) & 1) << (2 * b)
    return r


def decompose(w):  # (even bits, odd bits shifted into even positions)
    e = w & 0x55
    od = (w >> 1) & 0x55
    return e, od


# column numbering of the witness
ADV = ["s_and", "s_xor", "s_or", "a", "a_e", "a_o", "b", "b_e", "b_o", "es", "es_e", "es_o", "os", "os_e", "os_o", "res"]
A = {name: expr.Advice(i) for i, name in enumerate(ADV)}
S_TABLE = expr.Selector(0)
T_EVEN = ("fixed", 1)
DECOMPOSED = [("a", "a_e", "a_o"), ("b", "b_e", "b_o"), ("es", "es_e", "es_o"), ("os", "os_e", "os_o")]


def constraint_system():
    s = S_TABLE * (A["s_and"] + A["s_xor"] + A["s_or"])
    two = expr.Constant(2)
    gates, lookups = [], []
    for word, even, odd in DECOMPOSED:
        gates.append(s * (A[even] + two * A[odd] - A[word]))
        lookups.append(s * A[even])
        lookups.append(s * A[odd])
    gates.append(s * (A["a_e"] + A["b_e"] - A["es"]))
    gates.append(s * (A["a_o"] + A["b_o"] - A["os"]))
    and_ = A["es_o"] + two * A["os_o"]
    xor = A["es_e"] + two * A["os_e"]
    gates.append(S_TABLE * A["s_and"] * (and_ - A["res"]))
    gates.append(S_TABLE * A["s_xor"] * (xor - A["res"]))
    gates.append(S_TABLE * A["s_or"] * (xor + and_ - A["res"]))
    return gates, lookups


def witness(n, used, seed):
    rng = random.Random(seed)
    cols = {name: [0] * n for name in ADV}
    s_table = [0] * n
    for row in range(used):
        s_table[row] = 1
        op = rng.choice(("and", "xor", "or", "none"))
        a, b = rng.randrange(1 << WORD_BITS), rng.randrange(1 << WORD_BITS)
        if row < 4:
            a, b = [(0, 0), (255, 255), (0xAA, 0x55), (255, 0)][row]
        if op == "none":          # table row without a logic instruction: every constraint is switched off, cells arbitrary
            cols["a"][row], cols["res"][row] = a, rng.randrange(1 << 20)
            continue
        cols["s_" + op][row] = 1
        (ae, ao), (be, bo) = decompose(a), decompose(b)
        es, os_ = ae + be, ao + bo
        (ese, eso), (ose, oso) = decompose(es), decompose(os_)
        res = {"and": a & b, "xor": a ^ b, "or": a | b}[op]
        for name, v in (("a", a), ("a_e", ae), ("a_o", ao), ("b", b), ("b_e", be), ("b_o", bo), ("es", es), ("es_e", ese), ("es_o", eso),
                        ("os", os_), ("os_e", ose), ("os_o", oso), ("res", res)):
            cols[name][row] = v
    table = [spread_even(i) if i < (1 << (WORD_BITS // 2)) else 0 for i in range(n)]   # unassigned table rows read 0, which is in the table
    return cols, s_table, table


def to_dev(f, col):
    return torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda()


def from_dev(f, t):
    torch.cuda.synchronize()
    return [f.from_limbs(r) for r in t.cpu().numpy().view(np.uint64)]


def device_columns(f, cols, s_table, table):
    d = {("advice", i): to_dev(f, cols[name]) for i, name in enumerate(ADV)}
    d[("selector", 0)] = to_dev(f, s_table)
    d[T_EVEN] = to_dev(f, table)
    return d


def test_witness_satisfies_the_reference_semantics():
    """the witness builder itself: a.even + b.even etc. reproduce and / xor / or bit by bit (what the gates encode)"""
    cols, s_table, _ = witness(1 << K, 400, 0x106)
    seen = set()
    for row in range(400):
        for op, fn in (("and", lambda x, y: x & y), ("xor", lambda x, y: x ^ y), ("or", lambda x, y: x | y)):
            if cols["s_" + op][row]:
                seen.add(op)
                assert cols["res"][row] == fn(cols["a"][row], cols["b"][row])
    assert seen == {"and", "xor", "or"}


def test_gates_vanish_on_the_witness_and_catch_a_corrupted_cell():
    f = o.FIELDS[FIELD]
    n = 1 << K
    gates, _ = constraint_system()
    cols, s_table, table = witness(n, 400, 0x106)
    y = 0x7E57AB1E0DDC0FFEE % f.m
    ev = expr.GateEvaluator(expr.compile_gates(FIELD, gates, y))
    dev = device_columns(f, cols, s_table, table)
    used = {k: v for k, v in dev.items() if k in ev.program.columns}
    assert not any(from_dev(f, ev.eval(used, K, 1)))
    # the reference's negative tests: a wrong result / a wrong decomposition is caught on its own row, and only there
    for name, row in (("res", next(r for r in range(400) if cols["s_xor"][r])), ("es_o", next(r for r in range(400) if cols["s_and"][r])),
                      ("a_e", next(r for r in range(400) if cols["s_or"][r]))):
        bad = dict(cols)
        bad[name] = list(cols[name])
        bad[name][row] += 1
        dev_bad = device_columns(f, bad, s_table, table)
        h = from_dev(f, ev.eval({k: v for k, v in dev_bad.items() if k in ev.program.columns}, K, 1))
        assert [i for i, v in enumerate(h) if v] == [row], name
        # and the value is the oracle's
        want = o.evaluate_gates(f, [to_tuple(g) for g in gates], {k: [c[row]] for k, c in _int_columns(bad, s_table, table).items()}, y, 1)
        assert h[row] == want[0]


def _int_columns(cols, s_table, table):
    d = {("advice", i): cols[name] for i, name in enumerate(ADV)}
    d[("selector", 0)] = s_table
    d[T_EVEN] = table
    return d


def to_tuple(e):
    if isinstance(e, expr.Constant):
        return ("const", e.value)
    if isinstance(e, expr._Query):
        return ("col", (e.kind, e.column), e.rotation)
    if isinstance(e, expr.Negated):
        return ("neg", to_tuple(e.e))
    if isinstance(e, expr.Sum):
        return ("sum", to_tuple(e.a), to_tuple(e.b))
    if isinstance(e, expr.Product):
        return ("prod", to_tuple(e.a), to_tuple(e.b))
    if isinstance(e, expr.Scaled):
        return ("scaled", to_tuple(e.e), e.value)
    raise TypeError(e)


def test_quotient_of_the_logic_chip():
    f = o.FIELDS[FIELD]
    n = 1 << K
    gates, _ = constraint_system()
    assert max(g.degree() for g in gates) == 3
    cols, s_table, table = witness(n, 400, 0x107)
    dom = poly.EvaluationDomain(FIELD, 4, K)     # degree-3 gates: extended_k = K + 2
    y = 0x5EED5EED5EED5EED1234 % f.m
    prog = expr.compile_gates(FIELD, gates, y)
    dev = device_columns(f, cols, s_table, table)
    keys = list(prog.columns)
    lag = torch.stack([dev[k] for k in keys]).contiguous()
    coeff = dom.lagrange_to_coeff(lag.clone())
    ext = dom.coeff_to_extended(coeff)
    h = expr.GateEvaluator(prog).eval({k: ext[i] for i, k in enumerate(keys)}, dom.extended_k, 1 << (dom.extended_k - K)).reshape(1, -1, 4).contiguous()
    dom.divide_by_vanishing_poly(h)
    hc = dom.extended_to_coeff(h)[0]
    hc_host = from_dev(f, hc)
    assert any(hc_host[: 2 * n]) and not any(hc_host[2 * n:])
    x = 0x0F1E2D3C4B5A69788796A5B4C3D2E1F0 % f.m
    at = lambda t, pt: [f.from_limbs(r) for r in api.poly_eval_batch_dev(FIELD, t.contiguous(), t.shape[-2], t.shape[0], np.array(f.limbs(pt), np.uint64))]
    vals = dict(zip(keys, at(coeff, x)))
    folded = o.evaluate_gates(f, [to_tuple(g) for g in gates], {k: [v] for k, v in vals.items()}, y, 1)[0]   # no rotations in this chip
    assert at(hc.reshape(1, -1, 4), x)[0] * (pow(x, n, f.m) - 1) % f.m == folded


def test_even_bits_lookups_of_the_logic_chip():
    f = o.FIELDS[FIELD]
    n = 1 << K
    _, lookups = constraint_system()
    assert len(lookups) == 8
    cols, s_table, table = witness(n, 400, 0x108)
    dev = device_columns(f, cols, s_table, table)
    ints = _int_columns(cols, s_table, table)
    prog = expr.compile_outputs(FIELD, lookups)
    inputs = expr.GateEvaluator(prog, n_outputs=len(lookups)).eval({k: dev[k] for k in prog.columns}, K, 1)
    rng = random.Random(0x100C)
    beta, gamma = rng.randrange(f.m), rng.randrange(f.m)
    gp = permutation.lookup_product(FIELD, K, beta, gamma)
    for li, e in enumerate(lookups):
        want_in = [o.evaluate_expression(f, to_tuple(e), ints, r, n, 1) for r in range(n)]
        assert from_dev(f, inputs[li]) == want_in
        a_p, s_p = permutation.lookup_permute(FIELD, inputs[li].contiguous(), dev[T_EVEN])
        want_a, want_s = o.permute_expression_pair(want_in, table, n)
        got_a, got_s = from_dev(f, a_p), from_dev(f, s_p)
        assert got_a == want_a and got_s == want_s
        assert all(got_a[r] == got_s[r] or (r and got_a[r] == got_a[r - 1]) for r in range(n))
        z = from_dev(f, gp.compute({("advice", 0): inputs[li].contiguous(), ("advice", 1): dev[T_EVEN], ("advice", 2): a_p, ("advice", 3): s_p}))
        last = z[n - 1] * (want_in[n - 1] + beta) * (table[n - 1] + gamma) % f.m * pow((got_a[n - 1] + beta) * (got_s[n - 1] + gamma), -1, f.m) % f.m
        assert z[0] == 1 and last == 1


def test_lookup_rejects_a_word_part_with_odd_bits():
    f = o.FIELDS[FIELD]
    n = 1 << K
    _, lookups = constraint_system()
    cols, s_table, table = witness(n, 400, 0x109)
    row = next(r for r in range(400) if cols["s_and"][r])
    cols["a_e"][row] |= 2        # no longer an even-bits value (the decompose gate could still be satisfied by adjusting a_o)
    dev = device_columns(f, cols, s_table, table)
    prog = expr.compile_outputs(FIELD, [lookups[0]])
    inp = expr.GateEvaluator(prog, n_outputs=1).eval({k: dev[k] for k in prog.columns}, K, 1)
    with pytest.raises(api.TrhError):
        permutation.lookup_permute(FIELD, inp.contiguous(), dev[T_EVEN])


def test_quotient_of_the_sprod_gate():
    """the reference's highest-degree gate, /root/reference/src/circuits/sprod.rs:65-92 (degree 6: it is what makes the extended
    domain 8 n): s_table s_sprod (a_s b_s - d - 2^W c_s) with x_s = x_sigma (1 - 2 x_msb) the sign-magnitude value.  Witness:
    signed products of random words; the quotient h(X) must have degree < 5 n and satisfy h(x) (x^n - 1) = gate(x)."""
    f = o.FIELDS[FIELD]
    n = 1 << K
    W = WORD_BITS
    names = ["s_sprod", "a_sigma", "a_msb", "b_sigma", "b_msb", "c_sigma", "c_msb", "d"]
    q = {nm: expr.Advice(i) for i, nm in enumerate(names)}
    two, mx = expr.Constant(2), expr.Constant(1 << W)
    signed = lambda v: -q[v + "_msb"] * two * q[v + "_sigma"] + q[v + "_sigma"]
    gate = S_TABLE * q["s_sprod"] * (signed("a") * signed("b") - q["d"] - mx * signed("c"))
    assert gate.degree() == 6
    rng = random.Random(0x5B80D)
    cols = {nm: [0] * n for nm in names}
    s_table = [1 if r < 450 else 0 for r in range(n)]
    sign_mag = lambda v: (abs(v), 1 if v < 0 else 0)
    for r in range(450):
        if rng.randrange(4) == 0:
            cols["d"][r] = rng.randrange(1 << W)      # not an sprod row: unconstrained
            continue
        a, b = rng.randrange(-(1 << (W - 1)), 1 << (W - 1)), rng.randrange(-(1 << (W - 1)), 1 << (W - 1))
        p = a * b
        d = p % (1 << W)
        c = (p - d) >> W
        cols["s_sprod"][r] = 1
        (cols["a_sigma"][r], cols["a_msb"][r]), (cols["b_sigma"][r], cols["b_msb"][r]), (cols["c_sigma"][r], cols["c_msb"][r]) = sign_mag(a), sign_mag(b), sign_mag(c)
        cols["d"][r] = d
    dev = {("advice", i): to_dev(f, cols[nm]) for i, nm in enumerate(names)}
    dev[("selector", 0)] = to_dev(f, s_table)
    y = 1
    prog = expr.compile_gates(FIELD, [gate], y)
    keys = list(prog.columns)
    ev = expr.GateEvaluator(prog)
    assert not any(from_dev(f, ev.eval({k: dev[k] for k in keys}, K, 1)))            # satisfied on the rows
    dom = poly.EvaluationDomain(FIELD, 7, K)
    assert dom.extended_k == K + 3
    coeff = dom.lagrange_to_coeff(torch.stack([dev[k] for k in keys]).contiguous())
    ext = dom.coeff_to_extended(coeff)
    h = ev.eval({k: ext[i] for i, k in enumerate(keys)}, dom.extended_k, 1 << (dom.extended_k - K)).reshape(1, -1, 4).contiguous()
    dom.divide_by_vanishing_poly(h)
    hc = dom.extended_to_coeff(h)[0]
    hc_host = from_dev(f, hc)
    assert any(hc_host[4 * n: 5 * n]) and not any(hc_host[5 * n:])
    x = 0x1357924680ACE0BDF % f.m
    at = lambda t, pt: [f.from_limbs(r) for r in api.poly_eval_batch_dev(FIELD, t.contiguous(), t.shape[-2], t.shape[0], np.array(f.limbs(pt), np.uint64))]
    vals = dict(zip(keys, at(coeff, x)))
    want = o.evaluate_gates(f, [to_tuple(gate)], {k: [v] for k, v in vals.items()}, y, 1)[0]
    assert at(hc.reshape(1, -1, 4), x)[0] * (pow(x, n, f.m) - 1) % f.m == want
