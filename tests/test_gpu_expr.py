"""GPU parity of the gate-expression evaluator (csrc/expr.hip through the C ABI) against the oracle's restatement of
halo2's Expression::evaluate + y-folding (oracle/pasta.py::evaluate_gates).  Bit-exact."""
import ctypes
import random

import numpy as np
import pytest
import torch

import pasta as o
from tiny_ram_halo2_amd import api, expr, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init(0)
    yield


def to_tuple(e, key=lambda q: (q.kind, q.column)):
    if isinstance(e, expr.Constant):
        return ("const", e.value)
    if isinstance(e, expr._Query):
        return ("col", key(e), e.rotation)
    if isinstance(e, expr.Negated):
        return ("neg", to_tuple(e.e))
    if isinstance(e, expr.Sum):
        return ("sum", to_tuple(e.a), to_tuple(e.b))
    if isinstance(e, expr.Product):
        return ("prod", to_tuple(e.a), to_tuple(e.b))
    if isinstance(e, expr.Scaled):
        return ("scaled", to_tuple(e.e), e.value)
    raise TypeError(e)


def make_columns(f, keys, n, seed):
    rng = random.Random(seed)
    ints = {k: [rng.randrange(f.m) for _ in range(n)] for k in keys}
    # selectors are 0/1 columns, some values are small as witness cells are
    for k in keys:
        if k[0] == "selector":
            ints[k] = [rng.randrange(2) for _ in range(n)]
    dev = {k: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda() for k, col in ints.items()}
    return ints, dev


def from_dev(f, t):
    torch.cuda.synchronize()
    return [f.from_limbs(r) for r in t.cpu().numpy().view(np.uint64)]


@pytest.mark.parametrize("field", ["fp", "fq"])
@pytest.mark.parametrize("log_n,rot_step", [(4, 1), (9, 8), (6, 2)])
def test_synthetic_gates_vs_oracle(field, log_n, rot_step):
    f = o.FIELDS[field]
    n = 1 << log_n
    gates = expr.synthetic_gates(n_advice=12, n_fixed=3, n_gates=17, seed=log_n * 7 + rot_step)
    y = 0x1234567890ABCDEF1234567890ABCDEF % f.m
    prog = expr.compile_gates(field, gates, y)
    assert prog.max_degree == 6
    ints, dev = make_columns(f, prog.columns, n, seed=log_n)
    ev = expr.GateEvaluator(prog)
    got = from_dev(f, ev.eval(dev, log_n, rot_step))
    want = o.evaluate_gates(f, [to_tuple(g) for g in gates], ints, y, n, rot_step)
    assert got == want
    # a new proof, a new challenge: only the constant changes
    y2 = (y * y + 17) % f.m
    ev.set_challenge(y2)
    assert from_dev(f, ev.eval(dev, log_n, rot_step)) == o.evaluate_gates(f, [to_tuple(g) for g in gates], ints, y2, n, rot_step)


def test_deep_expression_spills_to_lds():
    """a balanced tree of depth 6 over two-entry leaves needs 8 stack entries: six of them live in LDS"""
    field, log_n = "fp", 5
    f = o.FIELDS[field]
    n = 1 << log_n

    def tree(d, i=0):
        if d == 0:
            return expr.Advice(i % 9, (i % 3) - 1) + (i + 1)
        return tree(d - 1, 2 * i) * tree(d - 1, 2 * i + 1) - expr.Fixed(i % 2, 0)

    g = tree(6)
    prog = expr.compile_gates(field, [g, -g * 3], 7)
    ints, dev = make_columns(f, prog.columns, n, seed=99)
    ev = expr.GateEvaluator(prog)
    assert expr._need(g) == 8 and ev.lds_slots() == 6
    assert from_dev(f, ev.eval(dev, log_n)) == o.evaluate_gates(f, [to_tuple(g), to_tuple(-g * 3)], ints, 7, n)


def test_edge_values_and_operand_order():
    """0, 1, m - 1 through every operator; SUB operand order; squaring of a shared node; rotation wrap-around"""
    field, log_n = "fq", 3
    f = o.FIELDS[field]
    n = 1 << log_n
    a, b = expr.Advice(0, 0), expr.Advice(1, -3)
    gates = [a - b, b - a, a * a, (a + b) * (a - b), -(a * 5) + b * (f.m - 1), expr.Constant(0) * a + 1, a * b * expr.Advice(1, 5)]
    vals = [0, 1, f.m - 1, 2, f.m - 2, (f.m - 1) // 2, 12345, 1 << 200]
    ints = {("advice", 0): vals, ("advice", 1): vals[::-1]}
    dev = {k: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda() for k, col in ints.items()}
    for y in (0, 1, f.m - 1):
        prog = expr.compile_gates(field, gates, y)
        got = from_dev(f, expr.GateEvaluator(prog).eval(dev, log_n))
        assert got == o.evaluate_gates(f, [to_tuple(g) for g in gates], ints, y, n)


def test_long_sums_force_inserted_reductions():
    """the machine's values are lazy (csrc/expr.hip): a loaded column is bounded by 32 m, sums add up, and trh_expr_create inserts a
    reduction where a sum could pass 250 m or a stored value 15 m.  Sums of 9 .. 40 raw columns at the extreme values m - 1 / 0 / 1,
    alone (stored directly), negated, squared, multiplied with each other, folded -- all bit-exact against the oracle"""
    field, log_n = "fp", 3
    f = o.FIELDS[field]
    n = 1 << log_n

    def total(k, start=0):
        e = expr.Advice(start % 6, 0)
        for i in range(1, k):
            e = e + expr.Advice((start + i) % 6, (i % 3) - 1)
        return e

    def alternating(k):
        e = expr.Advice(0, 0)
        for i in range(1, k):
            e = (e - expr.Advice(i % 6, 0)) if i % 2 else (e + expr.Advice(i % 6, 1))
        return e

    gates = [total(9), total(17, 2), -total(40, 1), total(12) * total(13, 3), total(20) * total(20) , alternating(33),
             (total(9) + 5) * (alternating(15) - 7) - total(8), total(40) * 3 + total(39)]
    vals = [f.m - 1, f.m - 1, 0, 1, f.m - 2, (f.m - 1) // 2, f.m - 1, 1 << 253]
    ints = {("advice", c): [vals[(r + c) % len(vals)] if c % 2 else f.m - 1 for r in range(n)] for c in range(6)}
    dev = {k: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda() for k, col in ints.items()}
    for y in (1, f.m - 1, 0x1234567):
        prog = expr.compile_gates(field, gates, y)
        assert from_dev(f, expr.GateEvaluator(prog).eval(dev, log_n)) == o.evaluate_gates(f, [to_tuple(g) for g in gates], ints, y, n)
    # one output per expression: every sum is STORED as it is (no multiplication behind it)
    prog = expr.compile_outputs(field, gates[:3])
    out = expr.GateEvaluator(prog, n_outputs=3).eval(dev, log_n)
    for i in range(3):
        want = [o.evaluate_expression(f, to_tuple(gates[i]), ints, r, n, 1) for r in range(n)]
        assert from_dev(f, out[i]) == want, i


def test_products_of_wide_sums_stay_inside_the_limbs():
    """a value B m has the top limb B 2^22, so the operands and the result of a multiplication have to stay below 512 m
    and the operand of a squaring below 256 m; trh_expr_create now bounds MUL / SQR operands as well.  S = c0 + .. + c6 with every
    column m - 1 (bound 224 m): (S S) S, its square, the square of S S (393 m), cubes and products of such products"""
    field, log_n = "fp", 3
    f = o.FIELDS[field]
    n = 1 << log_n
    S = expr.Advice(0, 0)
    for i in range(1, 7):
        S = S + expr.Advice(i, 0)
    P = S * S
    Q = P * S
    T = expr.Advice(0, 1)
    for i in range(1, 7):
        T = T - expr.Advice(i, 0)          # about -6 (m - 1): the negative side of the signed domain
    gates = [Q, Q * Q, P * P, (P * P) * (P * P), P * P * P, (S * T) * (T * T), (P + S) * (Q - T), ((S * S) * (T * T)) * (S + T), -(Q * Q) + P]
    for fill in (f.m - 1, 1, (f.m - 1) // 2):
        ints = {("advice", c): [fill if (r + c) % 5 else f.m - 1 for r in range(n)] for c in range(7)}
        dev = {k: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda() for k, col in ints.items()}
        for y in (1, 0x1234567):
            prog = expr.compile_gates(field, gates, y)
            assert from_dev(f, expr.GateEvaluator(prog).eval(dev, log_n)) == o.evaluate_gates(f, [to_tuple(g) for g in gates], ints, y, n)
        prog = expr.compile_outputs(field, gates[:4])
        out = expr.GateEvaluator(prog, n_outputs=4).eval(dev, log_n)
        for i in range(4):
            assert from_dev(f, out[i]) == [o.evaluate_expression(f, to_tuple(gates[i]), ints, r, n, 1) for r in range(n)], i


def test_program_validation():
    I = expr._Insn
    one = np.zeros((1, 4), np.uint64)

    def create(insns, n_columns=1, n_consts=1, n_outputs=1, n_locals=0):
        arr = (I * len(insns))(*[I(*t) for t in insns])
        h = ctypes.c_void_p()
        return api.lib().trh_expr_create(0, ctypes.cast(arr, ctypes.c_void_p), len(insns), api._p(one), n_consts, n_columns, n_outputs, n_locals, ctypes.byref(h))

    OP = expr.OP
    assert create([(OP["PUSH_COLUMN"], 0, 0), (OP["STORE_TOP"], 0, 0)]) == 0
    assert create([(OP["ADD"], 0, 0)]) != 0                                  # stack underflow
    assert create([(OP["PUSH_COLUMN"], 1, 0), (OP["STORE_TOP"], 0, 0)]) != 0  # column out of range
    assert create([(OP["PUSH_CONST"], 3, 0), (OP["STORE_TOP"], 0, 0)]) != 0   # constant out of range
    assert create([(OP["PUSH_COLUMN"], 0, 0), (OP["STORE_TOP"], 2, 0)]) != 0  # output out of range
    assert create([(99, 0, 0)]) != 0                                         # unknown opcode
    assert create([(OP["PUSH_LOCAL"], 0, 0), (OP["STORE_TOP"], 0, 0)]) != 0   # no locals declared
    assert "instruction" in api.lib().trh_last_error().decode()


def test_locals_share_a_subexpression():
    field, log_n = "fp", 4
    f = o.FIELDS[field]
    n = 1 << log_n
    OP, I = expr.OP, expr._Insn
    # t = a * b (kept in local 0); out0 = t + a; out1 = t * t
    insns = [(OP["PUSH_COLUMN"], 0, 0), (OP["PUSH_COLUMN"], 1, 1), (OP["MUL"], 0, 0), (OP["STORE_LOCAL"], 0, 0), (OP["PUSH_COLUMN"], 0, 0), (OP["ADD"], 0, 0),
             (OP["STORE_TOP"], 0, 0), (OP["PUSH_LOCAL"], 0, 0), (OP["SQR"], 0, 0), (OP["STORE_TOP"], 1, 0)]
    arr = (I * len(insns))(*[I(*t) for t in insns])
    h = ctypes.c_void_p()
    api._check(api.lib().trh_expr_create(api.FIELD_ID[field], ctypes.cast(arr, ctypes.c_void_p), len(insns), None, 0, 2, 2, 1, ctypes.byref(h)))
    ints, dev = make_columns(f, [("advice", 0), ("advice", 1)], n, seed=5)
    cols = (ctypes.c_void_p * 2)(dev[("advice", 0)].data_ptr(), dev[("advice", 1)].data_ptr())
    o0, o1 = torch.empty((n, 4), dtype=torch.int64, device="cuda"), torch.empty((n, 4), dtype=torch.int64, device="cuda")
    outs = (ctypes.c_void_p * 2)(o0.data_ptr(), o1.data_ptr())
    api._check(api.lib().trh_expr_eval_dev(h, cols, outs, log_n, 1, None))
    a, b = ints[("advice", 0)], ints[("advice", 1)]
    t = [a[i] * b[(i + 1) % n] % f.m for i in range(n)]
    assert from_dev(f, o0) == [(t[i] + a[i]) % f.m for i in range(n)]
    assert from_dev(f, o1) == [t[i] * t[i] % f.m for i in range(n)]
    api.lib().trh_expr_destroy(h)


@pytest.mark.parametrize("field", ["fp", "fq"])
def test_permutation_product_column(field):
    """plonk/permutation/prover.rs: z[i+1] = z[i] * prod_j (v_j + beta delta^j omega^i + gamma) / (v_j + beta sigma_j + gamma)"""
    from tiny_ram_halo2_amd import permutation
    f = o.FIELDS[field]
    k, ncol, first = 7, 4, 8  # the third chunk of a 4-columns-per-product permutation
    n = 1 << k
    rng = random.Random(0xBEE)
    vals = [[rng.randrange(f.m) for _ in range(n)] for _ in range(ncol)]
    sigs = [[rng.randrange(f.m) for _ in range(n)] for _ in range(ncol)]
    vals[1][5] = 0; sigs[2][9] = 0
    beta, gamma, z0 = rng.randrange(f.m), rng.randrange(f.m), rng.randrange(1, f.m)
    dev = lambda col: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda()
    pc = permutation.ProductColumn(field, k, ncol, first_column=first)
    z = from_dev(f, pc.compute([dev(c) for c in vals], [dev(c) for c in sigs], beta, gamma, z0))
    delta = pow(5, 1 << 32, f.m)
    assert delta == permutation.delta(field) and pow(delta, (f.m - 1) >> 32, f.m) == 1  # order divides t = (m - 1) / 2^32
    w = f.omega(k)
    want, acc = [], z0
    for i in range(n):
        want.append(acc)
        num = den = 1
        for j in range(ncol):
            num = num * (vals[j][i] + beta * pow(delta, first + j, f.m) * pow(w, i, f.m) + gamma) % f.m
            den = den * (vals[j][i] + beta * sigs[j][i] + gamma) % f.m
        acc = acc * num * pow(den, -1, f.m) % f.m
    assert z == want
    # a permutation that fixes every cell (sigma_j = delta^j omega^i): the product telescopes to z0 everywhere
    ident = [[pow(delta, first + j, f.m) * pow(w, i, f.m) % f.m for i in range(n)] for j in range(ncol)]
    z = from_dev(f, pc.compute([dev(c) for c in vals], [dev(c) for c in ident], beta, gamma, z0))
    assert z == [z0] * n


def test_lookup_product_column():
    """plonk/lookup/prover.rs commit_product: z[i+1] = z[i] (a_i + beta)(s_i + gamma) / ((a'_i + beta)(s'_i + gamma)); with
    a' a permutation of a and s' of s the product returns to 1 after the last row"""
    from tiny_ram_halo2_amd import permutation
    field, k = "fp", 8
    f = o.FIELDS[field]
    n = 1 << k
    rng = random.Random(0x100C)
    table = [rng.randrange(f.m) for _ in range(n)]
    a = [table[rng.randrange(n)] for _ in range(n)]
    ap, sp = sorted(a), list(table)
    rng.shuffle(sp)
    beta, gamma = rng.randrange(f.m), rng.randrange(f.m)
    dev = lambda col: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda()
    cols = {("advice", i): dev(c) for i, c in enumerate((a, table, ap, sp))}
    z = from_dev(f, permutation.lookup_product(field, k, beta, gamma).compute(cols))
    want, acc = [], 1
    for i in range(n):
        want.append(acc)
        acc = acc * (a[i] + beta) * (table[i] + gamma) % f.m * pow((ap[i] + beta) * (sp[i] + gamma), -1, f.m) % f.m
    assert z == want and acc == 1


def test_random_expression_trees():
    """every node type of halo2's Expression in random shapes (unbalanced trees exercise both operand orders of the lowering)"""
    field, log_n = "fp", 5
    f = o.FIELDS[field]
    n = 1 << log_n
    rng = random.Random(0x7EE5)

    def tree(depth):
        if depth == 0 or rng.random() < 0.15:
            kind = rng.randrange(5)
            if kind == 0:
                return expr.Constant(rng.choice([0, 1, f.m - 1, rng.randrange(f.m)]))
            cls = (expr.Advice, expr.Fixed, expr.Instance, expr.Selector)[kind - 1]
            return cls(rng.randrange(3), 0 if cls is expr.Selector else rng.randrange(-2, 3))
        kind = rng.randrange(5)
        if kind == 0:
            return expr.Negated(tree(depth - 1))
        if kind == 1:
            return expr.Scaled(tree(depth - 1), rng.randrange(f.m))
        a, b = tree(depth - 1), tree(rng.randrange(depth))
        if rng.random() < 0.5:
            a, b = b, a
        return expr.Sum(a, b) if kind in (2, 3) else expr.Product(a, b)

    for trial in range(12):
        gates = [tree(rng.randrange(1, 7)) for _ in range(rng.randrange(1, 6))]
        y = rng.randrange(f.m)
        prog = expr.compile_gates(field, gates, y)
        ints, dev = make_columns(f, prog.columns or [("advice", 0)], n, seed=trial)
        if not prog.columns:  # constant-only gates still need one column to size the launch
            continue
        got = from_dev(f, expr.GateEvaluator(prog).eval(dev, log_n, 2))
        assert got == o.evaluate_gates(f, [to_tuple(g) for g in gates], ints, y, n, 2), trial


@pytest.mark.parametrize("field", ["fp", "fq"])
def test_lookup_permuted_columns(field):
    """plonk/lookup/prover.rs permute_expression_pair: bit-exact against the step-by-step restatement (oracle), incl. the order in
    which the left-over table values fill the repeated rows; small values (high limbs zero), wide values, duplicates in the table"""
    from tiny_ram_halo2_amd import permutation
    f = o.FIELDS[field]
    rng = random.Random(0x100CA)
    dev = lambda col: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda()
    cases = []
    for n, tsize, wide in ((1, 1, False), (7, 3, False), (64, 16, True), (1000, 256, False), (5000, 700, True), (1 << 14, 1 << 12, True)):
        distinct = [rng.randrange(f.m) if wide else rng.randrange(1 << 16) for _ in range(tsize)]
        if wide and tsize > 4:
            distinct[0] = 0; distinct[1] = f.m - 1; distinct[2] = 1 << 64; distinct[3] = (1 << 64) - 1  # limb boundaries
        table = [distinct[i % tsize] for i in range(n)]          # duplicates in the table when tsize < n
        rng.shuffle(table)
        inp = [rng.choice(distinct) for _ in range(n)]
        cases.append((inp, table))
    inp = [5] * 300
    cases.append((inp, [5] + [9] * 299))   # one run only: every other row takes a left-over value
    for inp, table in cases:
        n = len(inp)
        want_a, want_s = o.permute_expression_pair(inp, table, n)
        a, s = permutation.lookup_permute(field, dev(inp), dev(table))
        assert from_dev(f, a) == want_a and from_dev(f, s) == want_s, n
        # the defining constraints of the argument
        for row in range(n):
            assert want_a[row] == want_s[row] or (row > 0 and want_a[row] == want_a[row - 1])
    # usable_rows < n: only the first rows take part
    inp, table = cases[3]
    want_a, want_s = o.permute_expression_pair(inp, table, 900)
    a, s = permutation.lookup_permute(field, dev(inp), dev(table), 900)
    assert from_dev(f, a) == want_a and from_dev(f, s) == want_s
    # an input value that is not in the table
    with pytest.raises(api.TrhError):
        permutation.lookup_permute(field, dev([1, 2, 3, 4]), dev([1, 2, 3, 3]))


@pytest.mark.parametrize("field", ["fp", "fq"])
def test_lookup_permuted_columns_batch(field):
    """trh_lookup_permute_batch_dev: several lookups of different character in ONE call (a 16-bit range table, full-size values, a
    single run, values that share their top limb or all but its low bits -- the ties that send a lookup through the all-limbs sort --, a
    constant column),
    usable_rows below the column length, every lookup against the oracle; an input missing from its table names the lookup"""
    from tiny_ram_halo2_amd import permutation
    f = o.FIELDS[field]
    rng = random.Random(0xBA7C4)
    rows, n = 5000, 4700   # ragged: not a multiple of the sort tile, usable rows below the column length
    def column_pair(kind):
        if kind == "range":
            distinct = [rng.randrange(1 << 16) for _ in range(300)]
        elif kind == "wide":
            distinct = [rng.randrange(f.m) for _ in range(700)] + [0, f.m - 1, 1 << 64, (1 << 64) - 1, 1 << 128, (1 << 192) + 5]
        elif kind == "ties":   # same top limb, different low limbs: the fast sort cannot order these
            top = rng.randrange(1 << 60) << 192
            distinct = [top + rng.randrange(1 << 190) for _ in range(50)] + [top + (v << 64) + 7 for v in range(20)] + [top + 7 + v for v in range(20)]
        elif kind == "low-ties":  # spread-out values, some of which differ only in the LOW bits of the top limb -- below the bits the fast
            distinct = [rng.randrange(f.m) for _ in range(600)]  # sort goes by (FAST_KEY_BITS): detected as ties, redone in the general form
            distinct += [v ^ (1 << 192) for v in distinct[:20]] + [v ^ (0x5A5 << 192) for v in distinct[20:40]] + [(v ^ (3 << 192)) + 1 for v in distinct[40:50]]
        elif kind == "one-run":
            distinct = [5]
        else:  # "constant": input constant, table with one more value
            distinct = [12345, 99]
        table = [distinct[i % len(distinct)] for i in range(n)]
        rng.shuffle(table)
        inp = [distinct[0]] * n if kind == "constant" else [rng.choice(distinct) for _ in range(n)]
        pad = [rng.randrange(f.m) for _ in range(rows - n)]   # rows behind usable_rows: ignored
        return inp + pad, table + pad[::-1]
    kinds = ["range", "wide", "ties", "one-run", "constant", "range", "ties", "wide", "low-ties"]
    pairs = [column_pair(kd) for kd in kinds]
    to_t = lambda cols: torch.from_numpy(np.array([[f.limbs(v) for v in c] for c in cols], dtype=np.uint64).view(np.int64)).cuda()
    a, s = permutation.lookup_permute_batch(field, to_t([p[0] for p in pairs]), to_t([p[1] for p in pairs]), n)
    torch.cuda.synchronize()
    for li, (inp, table) in enumerate(pairs):
        want_a, want_s = o.permute_expression_pair(inp, table, n)
        assert from_dev(f, a[li][:n]) == want_a and from_dev(f, s[li][:n]) == want_s, (li, kinds[li])
        assert not a[li][n:].any() and not s[li][n:].any()
    bad = [list(p) for p in pairs]
    bad[5] = ([1, 2, 3, 4] * (rows // 4), [1, 2, 3, 3] * (rows // 4))
    with pytest.raises(api.TrhError, match="lookup 5"):
        permutation.lookup_permute_batch(field, to_t([p[0] for p in bad]), to_t([p[1] for p in bad]), n)


@pytest.mark.parametrize("k", [9, 14])
def test_quotient_identity_end_to_end(k):
    """The chain create_proof runs for h(X), on a satisfied toy circuit, checked through an identity no single kernel can fake:
    columns a, b random and c = a * b on every row, gates s (a b - c) and s (c(X omega) - a(X omega) b(X omega)) folded with y.  Lagrange -> coefficients
    -> extended coset -> gate evaluation -> divide_by_vanishing_poly -> extended_to_coeff must give an h(X) with
        (1) zero coefficients from degree n on (the numerator has degree < 3n and is divisible by X^n - 1), and
        (2) h(x) (x^n - 1) == folded gates evaluated from the column polynomials at a random point x."""
    from tiny_ram_halo2_amd import poly
    field, j = "fp", 4   # degree-3 gates (selector x product): EvaluationDomain::new(4, k), extended_k = k + 2
    f = o.FIELDS[field]
    n = 1 << k
    dom = poly.EvaluationDomain(field, j, k)
    a_l = synth.field_elements(0xA0 + k, n)
    b_l = synth.field_elements(0xB0 + k, n)
    d_a, d_b = torch.from_numpy(a_l.view(np.int64)).cuda(), torch.from_numpy(b_l.view(np.int64)).cuda()
    d_c = torch.empty_like(d_a)
    api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS["mul"], api._devptr(d_a), api._devptr(d_b), api._devptr(d_c), n, None))
    one = np.array(f.limbs(1), np.uint64)
    d_s = torch.from_numpy(np.tile(one, (n, 1)).view(np.int64)).cuda()   # selector enabled on every row
    cols = torch.stack([d_a, d_b, d_c, d_s]).contiguous()                # Lagrange form, (4, n, 4)
    coeff = dom.lagrange_to_coeff(cols.clone())
    ext = dom.coeff_to_extended(coeff)
    A, B, C, S = expr.Advice(0), expr.Advice(1), expr.Advice(2), expr.Selector(0)
    An, Bn, Cn = expr.Advice(0, 1), expr.Advice(1, 1), expr.Advice(2, 1)
    gates = [S * (A * B - C), S * (Cn - An * Bn)]
    y = 0x1F2E3D4C5B6A7988 % f.m
    prog = expr.compile_gates(field, gates, y)
    res = {("advice", 0): ext[0], ("advice", 1): ext[1], ("advice", 2): ext[2], ("selector", 0): ext[3]}
    h = expr.GateEvaluator(prog).eval(res, dom.extended_k, 1 << (dom.extended_k - k)).reshape(1, -1, 4).contiguous()
    dom.divide_by_vanishing_poly(h)
    hc = dom.extended_to_coeff(h)[0]
    torch.cuda.synchronize()
    hc_host = hc.cpu().numpy().view(np.uint64)
    assert hc_host[: 2 * n].any() and not hc_host[2 * n:].any()     # (1): degree of h < 2n
    # (2) at a random point, from the coefficient forms
    x = 0x0123456789ABCDEF0FEDCBA987654321 % f.m
    w = f.omega(k)
    ev = lambda t, pt: [f.from_limbs(r) for r in api.poly_eval_batch_dev(field, t.contiguous(), t.shape[-2], t.shape[0], np.array(f.limbs(pt), np.uint64))]
    a_x, b_x, c_x, s_x = ev(coeff, x)
    a_wx, b_wx, c_wx, _ = ev(coeff, x * w % f.m)
    g0 = s_x * (a_x * b_x - c_x) % f.m
    g1 = s_x * (c_wx - a_wx * b_wx) % f.m
    folded = (g0 * y + g1) % f.m
    h_x = ev(hc.reshape(1, -1, 4), x)[0]
    assert h_x * (pow(x, n, f.m) - 1) % f.m == folded
    # the same chain over the coset-block layout with only the j - 1 blocks the quotient needs: rotations stay inside a block, the
    # vanishing polynomial is a constant per block, the quotient comes back from j - 1 inverse size-n transforms -- the SAME h(X)
    D = dom.quotient_poly_degree
    extb = dom.coeff_to_extended_blocks(coeff, D)                                     # (4, D, n, 4)
    resb = {("advice", 0): extb[0], ("advice", 1): extb[1], ("advice", 2): extb[2], ("selector", 0): extb[3]}
    hb = expr.GateEvaluator(prog).eval_blocks(resb, k, D).contiguous()
    hq = dom.blocks_to_quotient(hb)
    torch.cuda.synchronize()
    assert (hq.cpu().numpy().view(np.uint64) == hc_host[: D * n]).all()


def test_permutation_argument_closes_on_a_real_permutation():
    """copy constraints honoured by the witness: values constant on the cycles of a random permutation pi of all cells, sigma the
    encoding of pi (delta^j' omega^i' for pi(j, i) = (j', i')): the grand product over all rows is 1, i.e. z returns to z0"""
    from tiny_ram_halo2_amd import permutation
    field, k, ncol = "fq", 7, 4
    f = o.FIELDS[field]
    n = 1 << k
    rng = random.Random(0xC0B7)
    cells = [(j, i) for j in range(ncol) for i in range(n)]
    image = cells[:]
    rng.shuffle(image)
    pi = dict(zip(cells, image))
    vals = {}
    for c in cells:                      # one value per cycle
        if c in vals:
            continue
        v, cur = rng.randrange(f.m), c
        while cur not in vals:
            vals[cur] = v
            cur = pi[cur]
    delta, w = pow(5, 1 << 32, f.m), f.omega(k)
    label = lambda j, i: pow(delta, j, f.m) * pow(w, i, f.m) % f.m
    v_cols = [[vals[(j, i)] for i in range(n)] for j in range(ncol)]
    s_cols = [[label(*pi[(j, i)]) for i in range(n)] for j in range(ncol)]
    beta, gamma = rng.randrange(f.m), rng.randrange(f.m)
    dev = lambda col: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda()
    z = from_dev(f, permutation.ProductColumn(field, k, ncol).compute([dev(c) for c in v_cols], [dev(c) for c in s_cols], beta, gamma))
    num = den = 1
    for j in range(ncol):
        num = num * (v_cols[j][n - 1] + beta * label(j, n - 1) + gamma) % f.m
        den = den * (v_cols[j][n - 1] + beta * s_cols[j][n - 1] + gamma) % f.m
    assert z[0] == 1 and z[n - 1] * num % f.m == den      # z[n] = z[n-1] num / den = 1
    # and it does NOT close when one copy constraint is violated
    v_cols[2][5] = (v_cols[2][5] + 1) % f.m
    z = from_dev(f, permutation.ProductColumn(field, k, ncol).compute([dev(c) for c in v_cols], [dev(c) for c in s_cols], beta, gamma))
    num = den = 1
    for j in range(ncol):
        num = num * (v_cols[j][n - 1] + beta * label(j, n - 1) + gamma) % f.m
        den = den * (v_cols[j][n - 1] + beta * s_cols[j][n - 1] + gamma) % f.m
    assert z[n - 1] * num % f.m != den


def test_grand_products_batch_equals_single_columns():
    """the batched path (one inversion over every denominator, batched prefix product) gives the same z columns"""
    from tiny_ram_halo2_amd import permutation
    field, k, ncol, chunks = "fp", 9, 4, 5
    f = o.FIELDS[field]
    n = 1 << k
    rng = random.Random(0xBA7C)
    dev = lambda col: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda()
    beta, gamma = rng.randrange(f.m), rng.randrange(f.m)
    pcs, vals, sigs = [], [], []
    for c in range(chunks):
        pcs.append(permutation.ProductColumn(field, k, ncol, first_column=c * ncol))
        vals.append([dev([rng.randrange(f.m) for _ in range(n)]) for _ in range(ncol)])
        sigs.append([dev([rng.randrange(f.m) for _ in range(n)]) for _ in range(ncol)])
    singles = [from_dev(f, pcs[c].compute(vals[c], sigs[c], beta, gamma)) for c in range(chunks)]
    z = permutation.grand_products_batch(field, k, [pc.evaluator(beta, gamma) for pc in pcs], [pcs[c].columns(vals[c], sigs[c]) for c in range(chunks)])
    for c in range(chunks):
        assert from_dev(f, z[c]) == singles[c], c


@pytest.mark.parametrize("field", ["fp", "fq"])
def test_grand_products_terms_vs_big_ints(field):
    """the fixed-function form (trh_product_terms_dev + trh_field_batch_invert_mul_dev + batched prefix product): permutation chunks
    (a ragged last chunk, a zero denominator) and lookup products in ONE call against big-int arithmetic and the expression-program path"""
    from tiny_ram_halo2_amd import api, permutation
    f = o.FIELDS[field]
    k, ncol, n_columns, lookups = 9, 4, 10, 3  # chunks of 4, 4, 2 columns
    n = 1 << k
    rng = random.Random(0x7E2)
    dev = lambda col: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda()
    beta, gamma = rng.randrange(f.m), rng.randrange(f.m)
    vals = [[rng.randrange(f.m) for _ in range(n)] for _ in range(n_columns)]
    sigs = [[rng.randrange(f.m) for _ in range(n)] for _ in range(n_columns)]
    vals[5][17] = (-(beta * sigs[5][17] + gamma)) % f.m  # a zero denominator term: ff::BatchInvert leaves the zero, the ratio is 0
    d_vals, d_sigs = [dev(c) for c in vals], [dev(c) for c in sigs]
    w, delta = f.omega(k), permutation.delta(field)
    om = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    api.powers_dev(field, om, n, np.array(f.limbs(w), dtype=np.uint64))
    num_rows, den_rows, want = [], [], []
    for c0 in range(0, n_columns, ncol):
        c1 = min(c0 + ncol, n_columns)
        nr, dr = permutation.permutation_terms(field, d_vals[c0:c1], d_sigs[c0:c1], om, beta, gamma, first_column=c0)
        num_rows.append(nr); den_rows.append(dr)
        z, acc = [], 1
        for i in range(n):
            z.append(acc)
            num = den = 1
            for j in range(c0, c1):
                num = num * (vals[j][i] + beta * pow(delta, j, f.m) * pow(w, i, f.m) + gamma) % f.m
                den = den * (vals[j][i] + beta * sigs[j][i] + gamma) % f.m
            acc = acc * num * (pow(den, -1, f.m) if den else 0) % f.m
        want.append(z)
    for li in range(lookups):
        table = [rng.randrange(f.m) for _ in range(n)]
        a = [table[rng.randrange(n)] for _ in range(n)]
        ap, sp = sorted(a), list(table)
        rng.shuffle(sp)
        nr, dr = permutation.lookup_terms(field, dev(a), dev(table), dev(ap), dev(sp), beta, gamma)
        num_rows.append(nr); den_rows.append(dr)
        z, acc = [], 1
        for i in range(n):
            z.append(acc)
            acc = acc * (a[i] + beta) * (table[i] + gamma) % f.m * pow((ap[i] + beta) * (sp[i] + gamma), -1, f.m) % f.m
        assert acc == 1
        want.append(z)
    z = permutation.grand_products_terms(field, k, num_rows, den_rows)
    assert z.shape == (len(want), n, 4)
    for r in range(len(want)):
        assert from_dev(f, z[r]) == want[r], r
    # the expression-program path gives the same first chunk
    pc = permutation.ProductColumn(field, k, ncol, first_column=0)
    assert from_dev(f, pc.compute(d_vals[:ncol], d_sigs[:ncol], beta, gamma)) == want[0]


@pytest.mark.parametrize("n", [1, 63, 256, 257, 256 * 64, 256 * 64 + 1, 3 * 256 * 64 - 5])
def test_batch_invert_sizes_and_zeros(n):
    """ff::BatchInvert over the interleaved chunks: every size class of the last workgroup, zeros left alone, and the fused multiply"""
    from tiny_ram_halo2_amd import api
    field = "fp"
    f = o.FIELDS[field]
    rng = random.Random(n)
    a = [rng.randrange(1, f.m) for _ in range(n)]
    for z in range(0, n, 97):
        a[z] = 0
    b = [rng.randrange(f.m) for _ in range(n)]
    dev = lambda col: torch.from_numpy(np.array([f.limbs(v) for v in col], dtype=np.uint64).view(np.int64)).cuda()
    d = dev(a)
    api.batch_invert_dev(field, d, n)
    assert from_dev(f, d) == [pow(v, -1, f.m) if v else 0 for v in a]
    d = dev(a)
    api.batch_invert_mul_dev(field, d, dev(b), n)
    assert from_dev(f, d) == [pow(v, -1, f.m) * w % f.m if v else 0 for v, w in zip(a, b)]
