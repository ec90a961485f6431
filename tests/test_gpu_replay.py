"""GPU test (-m gpu): the create_proof schedule replay at k = 10 (WORD_BITS = 16, BASELINE config 1's
circuit size) with the first items of every primitive kind compared against the oracle."""
import numpy as np
import pytest

import cpu_ref
import pasta as o
from tiny_ram_halo2_amd import expr, replay

pytestmark = pytest.mark.gpu
import os
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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


@pytest.mark.parametrize("columns", ["random", "witness"])
def test_replay_k10_matches_oracle(columns):
    """`witness`: the value classes of the reference's tables (flags / words on n / 4 live rows, zero padding, blinding rows) --
    commitments whose pairs fall into a handful of buckets -- through the same oracle comparison"""
    seen = {}

    def hook(kind, inp, out):
        seen[kind] = seen.get(kind, 0) + 1
        if kind in ("commit_lagrange", "commit"):
            bases = inp["bases"].download()
            want = cpu_ref.to_affine("vesta", cpu_ref.best_multiexp("vesta", inp["scalars"], bases, threads=8))
            assert (np.asarray(out)[:8] == want).all(), kind
            return
        if kind == "lookup_permute":
            f = o.FIELDS[inp["field"]]
            a = [f.from_limbs(r) for r in inp["input"]]
            t = [f.from_limbs(r) for r in inp["table"]]
            want_a, want_s = o.permute_expression_pair(a, t, len(a))
            assert [f.from_limbs(r) for r in out[0]] == want_a and [f.from_limbs(r) for r in out[1]] == want_s
            return
        if kind == "product_column":  # plonk/permutation/prover.rs: the second chunk's z column against big ints
            f = o.FIELDS[inp["field"]]
            n = 1 << inp["k"]
            vals = [[f.from_limbs(r) for r in c] for c in inp["values"]]
            sigs = [[f.from_limbs(r) for r in c] for c in inp["sigmas"]]
            delta, w, beta, gamma = pow(5, 1 << 32, f.m), f.omega(inp["k"]), inp["beta"], inp["gamma"]
            acc, want = 1, []
            for i in range(n):
                want.append(acc)
                num = den = 1
                for j in range(4):
                    num = num * (vals[j][i] + beta * pow(delta, inp["first_column"] + j, f.m) * pow(w, i, f.m) + gamma) % f.m
                    den = den * (vals[j][i] + beta * sigs[j][i] + gamma) % f.m
                acc = acc * num * pow(den, -1, f.m) % f.m
            assert [f.from_limbs(r) for r in out] == want
            return
        if kind == "evals":  # arithmetic::eval_polynomial: Horner at the challenge
            f = o.FIELDS[inp["field"]]
            x, acc = f.from_limbs(inp["x"]), 0
            for r in np.asarray(inp["a"]).reshape(-1, 4)[::-1]:
                acc = (acc * x + f.from_limbs(r)) % f.m
            assert f.from_limbs(out) == acc
            return
        if kind == "h_eval" and "n_blocks" in inp:  # coset-block layout: a rotation stays inside its block
            f = o.FIELDS[inp["field"]]
            nb, D_ = 1 << inp["block_log"], inp["n_blocks"]
            got = out.cpu().numpy().view(np.uint64).reshape(D_, nb, 4)
            gates = [to_tuple(g) for g in inp["gates"]]
            for r, q in ((0, 0), (1, nb - 1), (D_ - 1, 4097 % nb), (2, nb // 2 + 3)):
                cols = Lazy({key: t.reshape(D_, nb, 4)[r] for key, t in inp["resident"].items()}, f)
                acc = 0
                for g in gates:
                    acc = (acc * inp["y"] + o.evaluate_expression(f, g, cols, q, nb, 1)) % f.m
                assert f.from_limbs(got[r, q]) == acc, (r, q)
            return
        if kind in ("coeff_to_extended_blocks", "blocks_to_quotient"):
            check_blocks(kind, inp, out)
            return
        if kind == "h_eval":  # sampled rows of the gate evaluation against the oracle's Expression::evaluate restatement
            f = o.FIELDS[inp["field"]]
            n = 1 << inp["log_n"]
            cols = {key: [f.from_limbs(r) for r in t.cpu().numpy().view(np.uint64)] for key, t in inp["resident"].items()}
            got = out.cpu().numpy().view(np.uint64)
            gates = [to_tuple(g) for g in inp["gates"]]
            for row in (0, 1, n - 1, 4097 % n, n // 2 + 3):
                acc = 0
                for g in gates:
                    acc = (acc * inp["y"] + o.evaluate_expression(f, g, cols, row, n, inp["rot_step"])) % f.m
                assert f.from_limbs(got[row]) == acc, row
            return
        field, j, k = inp["domain"]
        f = o.FIELDS[field]
        dom = o.EvaluationDomain(f, j, k)
        a = [f.from_limbs(r) for r in np.asarray(inp["a"]).reshape(-1, 4)]
        if kind == "lagrange_to_coeff":
            want = dom.lagrange_to_coeff(a)
        elif kind == "coeff_to_extended":
            want = dom.coeff_to_extended(a)
        else:
            want = dom.extended_to_coeff(dom.divide_by_vanishing_poly(a))
        assert [f.from_limbs(r) for r in np.asarray(out).reshape(-1, 4)] == want, kind

    res = replay.run(16, batch=32, hook=hook, verbose=False, columns=columns, gates_dir=GOLDEN)
    assert res["schedule"]["k"] == 10 and res["schedule"]["msm_n_plus_1"] == 504 and res["columns"] == columns
    assert res["counts"]["multiopen_folds"] == 4 and res["keygen_gpu_ms"]["columns"] == {"fixed": 25, "sigma": 188, "l0_l_blind_l_last": 3}
    assert res["counts"]["commit_lagrange"] == 497 and res["counts"]["coeff_to_extended"] == 497
    assert seen == {"lookup_permute": 1, "product_column": 1, "commit_lagrange": 9 if columns == "witness" else 3, "lagrange_to_coeff": 3, "coeff_to_extended_blocks": 3, "evals": 3, "h_eval": 2, "commit": 1,
                    "blocks_to_quotient": 1}
    assert res["h_eval_real_gates"]["gates"] == 142 and res["h_eval_real_gates"]["by_site"]["exe.rs temp-var / trace gates"] == 90 and res["h_eval_real_gates_ms"] > 0
    assert res["extended_domain"].startswith("5 of 8")


def test_replay_k10_full_extended_domain_matches_oracle():
    """the halo2 0.2.0 layout of the extended domain (all 2^extended_k points, natural order) stays available: --extended full"""
    seen = {}

    def hook(kind, inp, out):
        seen[kind] = seen.get(kind, 0) + 1
        if kind not in ("coeff_to_extended", "divide_and_extended_to_coeff") or seen[kind] > 1:
            return
        field, j, k = inp["domain"]
        dom = cpu_ref.EvaluationDomain(field, j, k)
        a = np.asarray(inp["a"]).reshape(-1, 4)
        want = dom.coeff_to_extended(a) if kind == "coeff_to_extended" else dom.extended_to_coeff(dom.divide_by_vanishing_poly(a))
        assert (np.asarray(out).reshape(-1, 4) == want).all(), kind

    res = replay.run(16, batch=32, hook=hook, verbose=False, columns="witness", keygen=False, extended="full", gates_dir=GOLDEN)
    assert res["extended_domain"] == "all 2^13 points" and seen["coeff_to_extended"] == 3 and seen["divide_and_extended_to_coeff"] == 1 and seen["h_eval"] == 2


def test_replay_k18_matches_oracle():
    """BASELINE config 4 (WORD_BITS = 32: k = 18, extended_k = 21, /root/reference/src/test_utils.rs:20, src/circuits/mod.rs:367):
    the same schedule (keygen section included) over WITNESS-SHAPED columns with the first item of every primitive kind -- and one
    commitment of every value class of the witness -- compared against the C++ oracle (oracle/cpu_ref.py:
    best_multiexp, EvaluationDomain over best_fft, eval_polynomial; the lookup permutation and the product column against the
    step-by-step big-int restatements)"""
    seen, classes = {}, []
    ftab = {"fp": o.FIELDS["fp"], "fq": o.FIELDS["fq"]}

    def hook(kind, inp, out):
        seen[kind] = seen.get(kind, 0) + 1
        if kind == "commit_lagrange" and "column_class" in inp:
            classes.append(inp["column_class"])
        elif seen[kind] > 1 and kind in ("commit_lagrange", "lagrange_to_coeff", "coeff_to_extended", "evals"):
            return  # the first item of each kind
        if kind in ("commit_lagrange", "commit"):
            want = cpu_ref.to_affine("vesta", cpu_ref.best_multiexp("vesta", inp["scalars"], inp["bases"].download(), threads=cpu_ref.hardware_threads()))
            assert (np.asarray(out)[:8] == want).all(), kind
            return
        if kind == "lookup_permute":
            f = ftab[inp["field"]]
            can = lambda a: synth_ints(cpu_ref.field_op(inp["field"], "from_mont", a))  # noqa: E731
            a, t = can(inp["input"]), can(inp["table"])
            want_a, want_s = o.permute_expression_pair(a, t, len(a))
            assert can(out[0]) == want_a and can(out[1]) == want_s
            del f
            return
        if kind == "product_column":
            field, k = inp["field"], inp["k"]
            f = ftab[field]
            n = 1 << k
            lim = lambda v: np.array(f.limbs(v), np.uint64)  # noqa: E731
            mul = lambda x, y: cpu_ref.field_op(field, "mul", x, y)  # noqa: E731
            add = lambda x, y: cpu_ref.field_op(field, "add", x, y)  # noqa: E731
            rep = lambda v: np.tile(lim(v), (n, 1))  # noqa: E731
            delta, w = pow(5, 1 << 32, f.m), f.omega(k)
            wp = np.tile(lim(1), (n, 1))  # omega^i by doubling
            span, cur = 1, w
            while span < n:
                wp[span:2 * span] = mul(wp[:span], np.tile(lim(cur), (span, 1)))
                cur, span = cur * cur % f.m, span * 2
            num = den = None
            for j in range(4):
                v, sg = inp["values"][j], inp["sigmas"][j]
                nj = add(add(v, mul(wp, rep(inp["beta"] * pow(delta, inp["first_column"] + j, f.m)))), rep(inp["gamma"]))
                dj = add(add(v, mul(sg, rep(inp["beta"]))), rep(inp["gamma"]))
                num = nj if num is None else mul(num, nj)
                den = dj if den is None else mul(den, dj)
            ratio = mul(num, cpu_ref.field_op(field, "inv", den))
            assert (np.asarray(out) == cpu_ref.prefix_product(field, ratio)).all()
            return
        if kind == "evals":
            assert (np.asarray(out) == cpu_ref.eval_polynomial(inp["field"], np.asarray(inp["a"]).reshape(-1, 4), inp["x"])).all()
            return
        if kind == "h_eval":  # sampled rows against the oracle's Expression::evaluate restatement (only the rows the gates touch are converted)
            f = ftab[inp["field"]]
            nb, D_ = 1 << inp["block_log"], inp["n_blocks"]   # coset-block layout: a rotation stays inside its block
            got = out.cpu().numpy().view(np.uint64).reshape(D_, nb, 4)
            gates = [to_tuple(g) for g in inp["gates"]]
            for r, q in ((0, 0), (1, nb - 1), (D_ - 1, 4097 % nb), (2, nb // 2 + 3), (3, 1)):
                cols = Lazy({key: t.reshape(D_, nb, 4)[r] for key, t in inp["resident"].items()}, f)
                acc = 0
                for g in gates:
                    acc = (acc * inp["y"] + o.evaluate_expression(f, g, cols, q, nb, 1)) % f.m
                assert f.from_limbs(got[r, q]) == acc, (r, q)
            return
        if kind in ("coeff_to_extended_blocks", "blocks_to_quotient"):
            check_blocks(kind, inp, out)
            return
        field, j, k = inp["domain"]
        dom = cpu_ref.EvaluationDomain(field, j, k)
        a = np.asarray(inp["a"]).reshape(-1, 4)
        if kind == "lagrange_to_coeff":
            want = dom.lagrange_to_coeff(a)
        elif kind == "coeff_to_extended":
            want = dom.coeff_to_extended(a)
        else:
            want = dom.extended_to_coeff(dom.divide_by_vanishing_poly(a))
        assert (np.asarray(out).reshape(-1, 4) == want).all(), kind

    res = replay.run(32, batch=32, hook=hook, verbose=False, columns="witness", keygen=True, gates_dir=GOLDEN)
    assert res["schedule"]["k"] == 18 and res["schedule"]["extended_k"] == 21 and res["schedule"]["msm_n_plus_1"] == 504
    # one commitment per value class beyond the first three columns: unblinded flags are the first columns, then
    # unblinded words, blinded flags / words / even-bits words, the sorted lookup columns and the full-size ones
    assert classes == [("word", False), ("flag", True), ("word", True), ("even", True), ("sorted", True), ("full", True)]
    assert res["keygen_gpu_ms"]["columns"] == {"fixed": 25, "sigma": 188, "l0_l_blind_l_last": 3}
    assert res["counts"]["commit_lagrange"] == 497 and res["counts"]["coeff_to_extended"] == 497 and res["counts"]["ipa"] == 1
    assert set(seen) == {"lookup_permute", "product_column", "commit_lagrange", "lagrange_to_coeff", "coeff_to_extended_blocks", "evals", "h_eval", "commit", "blocks_to_quotient"}


class Lazy(dict):
    """resident device columns read row by row (canonical ints): sampling a few rows of 2^18 .. 2^21 does not convert the columns"""

    def __init__(self, tensors, f):
        super().__init__()
        self.tensors, self.f = tensors, f

    def __missing__(self, key):
        col = LazyColumn(self.f, self.tensors[key])
        self[key] = col
        return col


def check_blocks(kind, inp, out):
    """the coset-block forms against EvaluationDomain's own functions (C++ oracle): block r, entry q of coeff_to_extended_blocks ==
    coeff_to_extended(a)[8 q + r]; blocks_to_quotient's polynomial, brought back to the extended coset by the oracle, takes the
    values input x (X^n - 1)^-1 on every point of the blocks it was interpolated from (the interpolant of degree < 5 n is unique)"""
    field, j, k = inp["domain"]
    f = o.FIELDS[field]
    dom = cpu_ref.EvaluationDomain(field, j, k)
    n, N, D_ = 1 << k, 1 << dom.extended_k, inp["n_blocks"]
    step = N // n
    if kind == "coeff_to_extended_blocks":
        want = np.asarray(dom.coeff_to_extended(np.asarray(inp["a"]).reshape(n, 4))).reshape(N, 4)
        got = np.asarray(out).reshape(D_, n, 4)
        for r in range(D_):
            assert (got[r] == want[r::step]).all(), r
        return
    h = np.asarray(out).reshape(D_ * n, 4)
    padded = np.zeros((N, 4), dtype=np.uint64)
    zs = [1, dom.c.g_coset, dom.c.g_coset_inv]
    fac = np.array([f.limbs(zs[i % 3]) for i in range(3)], dtype=np.uint64)
    padded[: D_ * n] = cpu_ref.field_op(field, "mul", h, np.tile(fac, (D_ * n // 3 + 1, 1))[: D_ * n])
    h_ext = cpu_ref.best_fft(field, padded, np.array(f.limbs(dom.c.extended_omega), np.uint64), dom.extended_k, cpu_ref.hardware_threads())
    num = np.asarray(inp["a"]).reshape(D_, n, 4)
    tinv = np.array([f.limbs(v) for v in dom.c.t_evaluations], dtype=np.uint64)
    for r in range(D_):
        assert (h_ext[r::step] == cpu_ref.field_op(field, "mul", num[r], np.tile(tinv[r % len(tinv)], (n, 1)))).all(), r


def synth_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


class LazyColumn:
    """a resident device column read row by row (canonical ints), so that sampling five rows of 2^21 does not convert the column"""

    def __init__(self, f, t):
        self.f, self.t = f, t

    def __getitem__(self, i):
        return self.f.from_limbs(self.t[i].cpu().numpy().view(np.uint64))
