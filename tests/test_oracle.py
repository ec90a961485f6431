"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors.

oracle/pasta.py (big-int) is pinned to the published pasta_curves constants; oracle/cpu_ref.cpp
(4 x u64 Montgomery restatement of best_multiexp / best_fft) is pinned to pasta.py through the
golden vectors.  The reference's own tests hold no value-level vectors for this path
(/root/reference/src/test_utils.rs:6-71 asserts only that the verifier accepts).
"""
import os

import numpy as np
import pytest

import cpu_ref
import pasta as o
from common import load_json, load_npz, unhex, unhex_rows

FIELDS = ["fp", "fq"]
CURVES = ["pallas", "vesta"]


def test_published_constants():
    o.check_published_constants()
    for c in o.CURVES.values():
        assert c.is_on_curve(c.generator)
        assert c.mul(c.scalar.m, c.generator) is None


@pytest.mark.parametrize("field", FIELDS)
def test_field_kat_cpu_ref(field):
    kat = load_json("field_kat.json")[field]
    f = o.FIELDS[field]
    a = unhex_rows([r["a"] for r in kat["rows"]])
    b = unhex_rows([r["b"] for r in kat["rows"]])
    for op in ("add", "sub", "mul"):
        got = cpu_ref.field_op(field, op, a, b)
        assert (got == unhex_rows([r[op] for r in kat["rows"]])).all(), op
    for op in ("sqr", "neg", "inv"):
        got = cpu_ref.field_op(field, op, a)
        assert (got == unhex_rows([r[op] for r in kat["rows"]])).all(), op
    # to_repr(): Montgomery -> canonical limbs and back
    can = cpu_ref.field_op(field, "from_mont", a)
    assert (can == unhex_rows([r["a_canonical"] for r in kat["rows"]])).all()
    assert (cpu_ref.field_op(field, "to_mont", can) == a).all()
    # the big-int oracle reproduces its own fixtures (guards against fixture rot)
    for r in kat["rows"][:8]:
        x, y = f.from_limbs(unhex(r["a"])), f.from_limbs(unhex(r["b"]))
        assert f.limbs(f.mul(x, y)) == [int(v) for v in unhex(r["mul"])]


@pytest.mark.parametrize("curve", CURVES)
def test_curve_kat_cpu_ref(curve):
    kat = load_json("curve_kat.json")[curve]
    for case in kat["add_cases"]:
        pj, qj = unhex(case["p_jac"]), unhex(case["q_jac"])
        want = unhex(case["sum_affine"])
        assert (cpu_ref.to_affine(curve, cpu_ref.point_op(curve, "add", pj, qj)) == want).all()
        assert (cpu_ref.to_affine(curve, cpu_ref.point_op(curve, "madd", pj, unhex(case["q_affine"]))) == want).all()
        assert (cpu_ref.to_affine(curve, cpu_ref.point_op(curve, "dbl", pj)) == unhex(case["dbl_p_affine"])).all()
        assert (cpu_ref.to_affine(curve, pj) == unhex(case["p_affine"])).all()
    for case in kat["mul_cases"]:
        got = cpu_ref.scalar_mul(curve, unhex(case["base"]), unhex(case["k_canonical"]))
        assert (cpu_ref.to_affine(curve, got) == unhex(case["result_affine"])).all()


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_msm_kat_cpu_ref(curve, threads):
    kat = load_json("msm_kat.json")[curve]
    for case in kat["cases"]:
        got = cpu_ref.best_multiexp(curve, unhex_rows(case["scalars"]), unhex_rows(case["bases"]), threads)
        assert (cpu_ref.to_affine(curve, got) == unhex(case["result_affine"])).all(), case["n"]


@pytest.mark.parametrize("curve", CURVES)
def test_msm_recipes_cpu_ref(curve):
    import importlib
    synth = importlib.import_module("tiny_ram_halo2_amd.synth")
    kat = load_json("msm_kat.json")[curve]
    for rec in kat["recipes"]:
        sc = synth.field_elements(rec["seed"], rec["n"])
        bases = cpu_ref.gen_bases(curve, rec["s0"], rec["d"], rec["n"], threads=4)
        got = cpu_ref.best_multiexp(curve, sc, bases, threads=4)
        assert (cpu_ref.to_affine(curve, got) == unhex(rec["result_affine"])).all(), rec["n"]


def test_msm_length_mismatch_asserts():
    # reference: assert_eq!(coeffs.len(), bases.len()) panics
    with pytest.raises(AssertionError):
        cpu_ref.best_multiexp("pallas", np.zeros((3, 4), np.uint64), np.zeros((2, 8), np.uint64))


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("threads", [1, 8])
def test_ntt_kat_cpu_ref(field, threads):
    arr, meta = load_npz("ntt_kat.npz"), load_json("ntt_kat.json")
    for log_n in (0, 1, 2, 3, 4, 10):
        key = f"{field}_{log_n}"
        fwd = cpu_ref.best_fft(field, arr[key + "_in"], unhex(meta[key]["omega"]), log_n, threads)
        assert (fwd == arr[key + "_fwd"].reshape(-1, 4)).all(), key
        inv = cpu_ref.best_fft(field, fwd, unhex(meta[key]["omega_inv"]), log_n, threads)
        assert (inv == arr[key + "_inv_unscaled"].reshape(-1, 4)).all(), key
        n = 1 << log_n
        ninv = np.tile(unhex(meta[key]["n_inv"]), (n, 1))
        assert (cpu_ref.field_op(field, "mul", inv, ninv) == arr[key + "_in"].reshape(-1, 4)).all()


@pytest.mark.parametrize("field", FIELDS)
def test_ntt_roundtrip_and_horner_2_16(field):
    """size-independent properties at a larger size: inverse(forward(a)) * n^-1 == a and
    a'[i] == poly(omega^i) by Horner at a few i (big-int)."""
    import importlib
    synth = importlib.import_module("tiny_ram_halo2_amd.synth")
    f = o.FIELDS[field]
    log_n = 16
    a = synth.ntt_input(log_n)
    w = f.omega(log_n)
    fwd = cpu_ref.best_fft(field, a, np.array(f.limbs(w), np.uint64), log_n, threads=8)
    coeffs = [f.from_limbs(r) for r in a]
    for i in (0, 1, 12345, (1 << log_n) - 1):
        x, acc = pow(w, i, f.m), 0
        for cf in reversed(coeffs):
            acc = (acc * x + cf) % f.m
        assert f.from_limbs(fwd[i]) == acc
    inv = cpu_ref.best_fft(field, fwd, np.array(f.limbs(f.inv(w)), np.uint64), log_n, threads=8)
    ninv = np.tile(np.array(f.limbs(f.inv(1 << log_n)), np.uint64), (1 << log_n, 1))
    assert (cpu_ref.field_op(field, "mul", inv, ninv) == a).all()


def test_ipa_prover_restatement_satisfies_the_verifier_equation():
    """the two oracle restatements (commitment::create_proof and commitment::verify_proof) agree: what the reference's own
    tests pin for this path is exactly `verifier accepts` (src/test_utils.rs:52-68)"""
    import random
    cv = o.CURVES["vesta"]
    fs = cv.scalar
    k, n = 3, 8
    rnd = random.Random(5)
    g = [cv.mul(rnd.randrange(1, fs.m), cv.generator) for _ in range(n)]
    w, u = cv.mul(12345, cv.generator), cv.mul(6789, cv.generator)

    class T:
        def __init__(self):
            self.pts, self.ch, self.sc, self.r = [], [], [], random.Random(9)

        def write_point(self, p):
            self.pts.append(p)

        def write_scalar(self, v):
            self.sc.append(v)

        def squeeze_challenge_scalar(self):
            c = self.r.randrange(1, fs.m)
            self.ch.append(c)
            return c

    p = [rnd.randrange(fs.m) for _ in range(n)]
    sp = [rnd.randrange(fs.m) for _ in range(n)]
    pb, sb, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
    t = T()
    c, f = o.ipa_create_proof(cv, k, g, w, u, lambda: rnd.randrange(fs.m), t, p, pb, x3, sp, sb)
    P = o.best_multiexp(cv, p + [pb], g + [w])
    v = 0
    for cf in reversed(p):
        v = (v * x3 + cf) % fs.m
    rounds = [(t.pts[1 + 2 * j], t.pts[2 + 2 * j]) for j in range(k)]
    assert o.ipa_verify_proof(cv, k, g, w, u, P, x3, v, t.pts[0], t.ch[0], t.ch[1], rounds, t.ch[2:], c, f)
    assert not o.ipa_verify_proof(cv, k, g, w, u, P, x3, (v + 1) % fs.m, t.pts[0], t.ch[0], t.ch[1], rounds, t.ch[2:], c, f)


def test_permutation_delta_is_the_published_constant():
    """pasta_curves 0.4.1 `DELTA` (fields/fp.rs, fields/fq.rs; crate pinned at /root/reference/Cargo.lock:847-858) = GENERATOR^(2^S)
    with GENERATOR = 5, S = 32: the constant the permutation argument's column labels delta^j omega^i are built from"""
    from tiny_ram_halo2_amd import permutation
    published = {
        "fp": [0x6A6CCD20DD7B9BA2, 0xF5E4F3F13EEE5636, 0xBD455B7112A5049D, 0x0A757D0F0006AB6C],
        "fq": [0x8494392472D1683C, 0xE3AC3376541D1140, 0x06F0A88E7F7949F8, 0x2237D54423724166],
    }
    for field, limbs in published.items():
        v = sum(w << (64 * i) for i, w in enumerate(limbs))
        assert permutation.delta(field) == v
        f = o.FIELDS[field]
        assert pow(v, (f.m - 1) >> 32, f.m) == 1 and pow(v, (f.m - 1) >> 33, f.m) != 1 or ((f.m - 1) >> 32) % 2 == 1


# ---- round-2 additions to the C++ oracle (IPA prover, curve-point FFT, hashed bases): pinned to the big-int restatements ----
from common import LimbTranscript, OracleTranscript  # noqa: E402


@pytest.mark.parametrize("curve", CURVES)
def test_cpp_hashed_bases_and_scaling(curve):
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 40
    pts = cpu_ref.gen_bases_hashed(curve, 0xABCDEF, n, threads=2)
    logs = cpu_ref.hashed_scalars(0xABCDEF, n)
    assert len({tuple(r) for r in pts.tolist()}) == n
    for i in (0, 1, 7, 39):
        h = o.limbs_to_int(logs[i])
        assert 0 < h < 1 << 254
        assert cv.affine_from_limbs(pts[i]) == cv.mul(h, cv.generator)
    ks = [3, fs.m - 1, 0, 0x1234567890ABCDEF1234567890ABCDEF % fs.m]
    got = cpu_ref.scale_points(curve, cv.affine_limbs(cv.generator), np.array([fs.limbs(k) for k in ks], np.uint64), threads=2)
    each = cpu_ref.scale_points_each(curve, pts[:4], np.array([fs.limbs(k) for k in ks], np.uint64), threads=2)
    for i, k in enumerate(ks):
        want = cv.mul(k, cv.generator)
        assert (cv.affine_from_limbs(got[i]) if got[i].any() else None) == want
        assert (cv.affine_from_limbs(each[i]) if each[i].any() else None) == cv.mul(k, cv.affine_from_limbs(pts[i]))


@pytest.mark.parametrize("curve,k", [("pallas", 3), ("vesta", 4)])
def test_cpp_point_fft_vs_bigint(curve, k):
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    g = cpu_ref.gen_bases_hashed(curve, 77 + k, n, threads=2)
    g[2] = 0  # an identity among the inputs
    omega = fs.inv(fs.omega(k))
    got = cpu_ref.best_fft_points(curve, g, np.array(fs.limbs(omega), np.uint64), k, threads=3)
    want = o.best_fft_points(cv, [cv.affine_from_limbs(r) if r.any() else None for r in g], omega, k)
    for i in range(n):
        assert (cv.affine_from_limbs(got[i]) if got[i].any() else None) == want[i], i


@pytest.mark.parametrize("curve,k", [("vesta", 3), ("pallas", 4)])
def test_cpp_ipa_prover_vs_bigint(curve, k):
    """oracle/cpu_ref.cpp::ipa_create_proof (used by the GPU parity tests at k = 10 .. 18) writes the transcript of
    oracle/pasta.py::ipa_create_proof, byte for byte"""
    import random
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    rnd = random.Random(0xC0DE + k)
    g_l = cpu_ref.gen_bases_hashed(curve, 5 + k, n, threads=2)
    w_l = cpu_ref.gen_bases(curve, 424242, 1, 1, threads=1)
    u_l = cpu_ref.gen_bases(curve, 737373, 1, 1, threads=1)
    p_poly = [rnd.randrange(fs.m) for _ in range(n)]
    s_poly = [rnd.randrange(fs.m) for _ in range(n)]
    p_blind, s_blind, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
    draws = [rnd.randrange(fs.m) for _ in range(2 * k)]
    it1, it2 = iter(draws), iter(draws)
    lim = lambda v: np.array(fs.limbs(v), np.uint64)  # noqa: E731
    t_cpp, t_ref = LimbTranscript(fs), OracleTranscript(cv)
    c_cpp, f_cpp = cpu_ref.ipa_create_proof(curve, k, g_l, w_l[0], u_l[0], lambda: lim(next(it1)), t_cpp, np.array([fs.limbs(v) for v in p_poly], np.uint64), lim(p_blind),
                                            lim(x3), np.array([fs.limbs(v) for v in s_poly], np.uint64), lim(s_blind), threads=2)
    c_ref, f_ref = o.ipa_create_proof(cv, k, [cv.affine_from_limbs(r) for r in g_l], cv.affine_from_limbs(w_l[0]), cv.affine_from_limbs(u_l[0]),
                                      lambda: next(it2), t_ref, p_poly, p_blind, x3, s_poly, s_blind)
    assert (fs.from_limbs(c_cpp), fs.from_limbs(f_cpp)) == (c_ref, f_ref)
    assert len(t_cpp.log) == 1 + 2 * k + 2
    assert t_cpp.log == t_ref.log


def test_fast_multiopen_and_verifier_vs_bigint():
    """tests/common.py's C++-backed multiopen driver and verifier equation (used at k >= 10 on the GPU box) against the big-int
    restatements, at k = 4"""
    import random
    from common import ipa_verify_fast, multiopen_create_proof_fast
    curve, k = "vesta", 4
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    rnd = random.Random(0x0BE1)
    g_l = cpu_ref.gen_bases_hashed(curve, 23, n, threads=2)
    w_l = cpu_ref.gen_bases(curve, 818181, 1, 1, threads=1)
    u_l = cpu_ref.gen_bases(curve, 929292, 1, 1, threads=1)
    keys = ["a", "b", "z"]
    polys = {key: [rnd.randrange(fs.m) for _ in range(n)] for key in keys}
    blinds = {key: rnd.randrange(fs.m) for key in keys}
    x = rnd.randrange(fs.m)
    xw = x * fs.omega(k) % fs.m
    queries = [(x, "a"), (x, "b"), (xw, "b"), (x, "z"), (xw, "z")]
    draws = [rnd.randrange(fs.m) for _ in range(2 + n + 1 + 2 * k)]
    it1, it2 = iter(draws), iter(draws)
    t_fast, t_ref = LimbTranscript(fs), OracleTranscript(cv)
    got = multiopen_create_proof_fast(curve, k, g_l, w_l[0], u_l[0], lambda: next(it1), t_fast, queries, polys, blinds)
    want = o.multiopen_create_proof(cv, k, [cv.affine_from_limbs(r) for r in g_l], cv.affine_from_limbs(w_l[0]), cv.affine_from_limbs(u_l[0]),
                                    lambda: next(it2), t_ref, queries, polys, blinds)
    assert got == want and t_fast.log == t_ref.log

    # verifier equation: an honest opening from the big-int prover is accepted by both forms, a tampered one rejected by both
    p_poly = [rnd.randrange(fs.m) for _ in range(n)]
    s_poly = [rnd.randrange(fs.m) for _ in range(n)]
    p_blind, s_blind, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
    rdraws = iter([rnd.randrange(fs.m) for _ in range(2 * k)])

    class Rec(OracleTranscript):
        def __init__(self, c):
            super().__init__(c)
            self.points, self.challenges = [], []

        def write_point(self, pt):
            self.points.append(pt)
            super().write_point(pt)

        def squeeze_challenge_scalar(self):
            c = super().squeeze_challenge_scalar()
            self.challenges.append(c)
            return c

    tr = Rec(cv)
    g = [cv.affine_from_limbs(r) for r in g_l]
    w, u = cv.affine_from_limbs(w_l[0]), cv.affine_from_limbs(u_l[0])
    c, f = o.ipa_create_proof(cv, k, g, w, u, lambda: next(rdraws), tr, p_poly, p_blind, x3, s_poly, s_blind)
    commitment = o.best_multiexp(cv, p_poly + [p_blind], g + [w])
    v = 0
    for cf in reversed(p_poly):
        v = (v * x3 + cf) % fs.m
    xi, z, ch = tr.challenges[0], tr.challenges[1], tr.challenges[2:]
    rounds = [(tr.points[1 + 2 * j], tr.points[2 + 2 * j]) for j in range(k)]
    assert o.ipa_verify_proof(cv, k, g, w, u, commitment, x3, v, tr.points[0], xi, z, rounds, ch, c, f)
    xy = lambda p: np.array(cv.affine_limbs(p), np.uint64)  # noqa: E731
    fast_args = (curve, k, g_l, w_l[0], u_l[0], xy(commitment), x3)
    fast_tail = (xy(tr.points[0]), xi, z, [(xy(a), xy(b)) for a, b in rounds], ch)
    assert ipa_verify_fast(*fast_args, v, *fast_tail, c, f)
    assert not ipa_verify_fast(*fast_args, (v + 1) % fs.m, *fast_tail, c, f)
    assert not ipa_verify_fast(*fast_args, v, *fast_tail, c, (f + 1) % fs.m)


def test_selftest_constants_are_what_the_oracle_generates(tmp_path):
    """csrc/selftest_kat.h (the known answers trh_init's self-test compares the device with) is DATA generated by
    tests/golden/make_selftest_kat.py from this oracle: regenerating it must reproduce the committed header byte for byte -- the library
    carries no constant the oracle does not vouch for, and a change of either side cannot slip through unnoticed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "selftest_kat.h"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_selftest_kat.py"), str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_text() == open(os.path.join(root, "tiny-ram-halo2_amd", "csrc", "selftest_kat.h")).read()
