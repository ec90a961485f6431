"""CPU test: the lowering of halo2 Expression trees to libtrh's stack program (tiny-ram-halo2_amd/expr.py) against the oracle's
Expression::evaluate restatement, through a big-int interpreter of the program (no device, no libtrh)."""
import random

import pasta as o
from tiny_ram_halo2_amd import expr


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
    return ("scaled", to_tuple(e.e), e.value)


def interpret(prog, f, columns, row, n, rot_step):
    names = {v: k for k, v in expr.OP.items()}
    st, acc, outs, depth_max = [], 0, {}, 0
    for op, a, rot in prog.insns:
        op = names[int(op)]
        if op == "PUSH_COLUMN":
            st.append(columns[prog.columns[a]][(row + rot * rot_step) % n])
        elif op == "PUSH_CONST":
            st.append(prog.consts[a])
        elif op in ("ADD", "SUB", "MUL"):
            t = st.pop()
            st[-1] = (st[-1] + t if op == "ADD" else st[-1] - t if op == "SUB" else st[-1] * t) % f.m
        elif op == "NEG":
            st[-1] = -st[-1] % f.m
        elif op == "SQR":
            st[-1] = st[-1] * st[-1] % f.m
        elif op == "MUL_CONST":
            st[-1] = st[-1] * prog.consts[a] % f.m
        elif op == "FOLD":
            acc = (acc * prog.consts[a] + st.pop()) % f.m
        elif op == "STORE_TOP":
            outs[a] = st.pop()
        elif op == "STORE_ACC":
            outs[a] = acc
        else:
            raise AssertionError(op)
        depth_max = max(depth_max, len(st))
    assert not st
    return outs, depth_max


def test_compile_gates_matches_expression_evaluate():
    f = o.FIELDS["fp"]
    rng = random.Random(0x10E4)
    n, rot_step = 16, 2
    gates = expr.synthetic_gates(n_advice=9, n_fixed=3, n_gates=40, seed=3)
    a, b = expr.Advice(0), expr.Advice(1, -1)
    gates += [a - b, b - a * 7, (a + b) * (a + b), -(a * b) + 3, expr.Constant(5) * 1 + a]
    y = rng.randrange(f.m)
    prog = expr.compile_gates("fp", gates, y)
    cols = {key: [rng.randrange(f.m) for _ in range(n)] for key in prog.columns}
    want = o.evaluate_gates(f, [to_tuple(g) for g in gates], cols, y, n, rot_step)
    for row in range(n):
        outs, depth = interpret(prog, f, cols, row, n, rot_step)
        assert outs[0] == want[row]
        assert depth <= max(expr._need(g) for g in gates)   # the Sethi-Ullman bound holds


def test_compile_outputs_and_synthetic_shape():
    f = o.FIELDS["fq"]
    rng = random.Random(1)
    num = (expr.Advice(0) + 5) * (expr.Advice(1) + 7)
    den = (expr.Advice(2) + 5) * (expr.Advice(3) + 7)
    prog = expr.compile_outputs("fq", [num, den])
    cols = {key: [rng.randrange(f.m) for _ in range(4)] for key in prog.columns}
    outs, _ = interpret(prog, f, cols, 1, 4, 1)
    c = lambda i: cols[("advice", i)][1]
    assert outs == {0: (c(0) + 5) * (c(1) + 7) % f.m, 1: (c(2) + 5) * (c(3) + 7) % f.m}
    assert max(g.degree() for g in expr.synthetic_gates(8, 2, 12)) == 6   # the reference's maximum constraint degree
