"""GPU parity tests (-m gpu) for the IPA opening: the device primitives
against the oracle, then the whole k-round prover replayed with an injected transcript and randomness
against the big-int restatement oracle/pasta.py::ipa_create_proof -- every L_j, R_j, c and f identical."""
import hashlib
import random

import numpy as np
import pytest
import torch

import cpu_ref
import pasta as o
from common import DeviceTranscript, HashTranscript, OracleTranscript
from tiny_ram_halo2_amd import api, ipa, poly, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init(0)
    yield


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64).copy()).cuda()


def to_host(t):
    torch.cuda.synchronize()
    return t.contiguous().cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("field", ["fp", "fq"])
def test_inner_product_axpy_powers(field):
    f = o.FIELDS[field]
    for n in (1, 7, 1000, 70000):
        a, b = synth.field_elements(0x1A, n), synth.field_elements(0x1B, n)
        da, db = to_dev(a), to_dev(b)
        got = api.inner_product_dev(field, da, db, n)
        prod = cpu_ref.field_op(field, "mul", a, b)
        want = 0
        for r in prod:
            want = (want + f.from_limbs(r)) % f.m
        assert f.from_limbs(got) == want
        c = np.array(f.limbs(0xC0FFEE + n), np.uint64)
        api.axpy_dev(field, da, db, n, c)
        wanty = cpu_ref.field_op(field, "add", a, cpu_ref.field_op(field, "mul", b, np.tile(c, (n, 1))))
        assert (to_host(da) == wanty).all()
    x = 0x123456789ABCDEF % f.m
    out = torch.empty((5000, 4), dtype=torch.int64, device="cuda")
    api.powers_dev(field, out, 5000, np.array(f.limbs(x), np.uint64))
    got = to_host(out)
    for i in (0, 1, 2, 63, 64, 4999):
        assert f.from_limbs(got[i]) == pow(x, i, f.m)


@pytest.mark.parametrize("curve", ["pallas", "vesta"])
def test_bases_fold(curve):
    cv = o.CURVES[curve]
    half = 300
    g = cpu_ref.gen_bases(curve, 17, 5, 2 * half, threads=4)
    g[3] = 0                       # identity in the low half
    g[half + 5] = 0                # identity in the high half
    g[half + 7] = g[7]             # lo == hi: lo + u * lo
    u = 0x1D0F5A7E9B3C2468ACE13579BDF02468ACE13579BDF0246 % cv.scalar.m
    d = to_dev(g)
    api.bases_fold_dev(curve, d[:half], d[half:], half, np.array(cv.scalar.limbs(u), np.uint64))
    got = to_host(d)[:half]
    for i in (0, 1, 3, 5, 7, 150, 299):
        want = cv.add(cv.affine_from_limbs(g[i]), cv.mul(u, cv.affine_from_limbs(g[half + i])))
        assert cv.affine_from_limbs(got[i]) == want, i
    # u = -1 on equal points cancels to the identity
    d2 = to_dev(np.concatenate([g[:half], g[:half]]))
    api.bases_fold_dev(curve, d2[:half], d2[half:], half, np.array(cv.scalar.limbs(cv.scalar.m - 1), np.uint64))
    assert (to_host(d2)[:half] == 0).all()


@pytest.mark.parametrize("curve,k", [("vesta", 3), ("vesta", 6), ("pallas", 5)])
def test_ipa_create_proof_vs_oracle(curve, k):
    cv = o.CURVES[curve]
    sfn = api.SCALAR_FIELD[curve]
    fs = cv.scalar
    n = 1 << k
    rnd = random.Random(0x1FA + k)
    g_l = cpu_ref.gen_bases(curve, 31, 7, n, threads=4)
    gl_l = cpu_ref.gen_bases(curve, 99991, 3, n, threads=4)
    w_l = cpu_ref.gen_bases(curve, 424242, 1, 1, threads=1)
    u_l = cpu_ref.gen_bases(curve, 737373, 1, 1, threads=1)
    params = poly.Params(curve, k, g_l, gl_l, w_l, u=u_l)
    p_poly = [rnd.randrange(fs.m) for _ in range(n)]
    s_poly = [rnd.randrange(fs.m) for _ in range(n)]
    p_blind, s_blind, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
    draws = [rnd.randrange(fs.m) for _ in range(2 * k)]

    it1, it2 = iter(draws), iter(draws)
    t_dev, t_ref = DeviceTranscript(fs.m), OracleTranscript(cv)
    c_dev, f_dev = ipa.create_proof(params, lambda: next(it1), t_dev, to_dev(np.array([fs.limbs(v) for v in p_poly], np.uint64)),
                                    p_blind, x3, s_poly=np.array([fs.limbs(v) for v in s_poly], np.uint64), s_blind=s_blind)
    c_ref, f_ref = o.ipa_create_proof(cv, k, [cv.affine_from_limbs(r) for r in g_l], cv.affine_from_limbs(w_l[0]),
                                      cv.affine_from_limbs(u_l[0]), lambda: next(it2), t_ref, p_poly, p_blind, x3, s_poly, s_blind)
    assert (c_dev, f_dev) == (c_ref, f_ref)
    assert len(t_dev.log) == len(t_ref.log) == 1 + 2 * k + 2
    for a, b in zip(t_dev.log, t_ref.log):
        assert a == b
    # the single-call C++ prover (trh_ipa_create_proof) writes the same transcript
    it3 = iter(draws)
    t_nat = DeviceTranscript(fs.m)
    c_nat, f_nat = ipa.create_proof_native(params, lambda: next(it3), t_nat, to_dev(np.array([fs.limbs(v) for v in p_poly], np.uint64)),
                                           p_blind, x3, np.array([fs.limbs(v) for v in s_poly], np.uint64), s_blind)
    assert (c_nat, f_nat) == (c_ref, f_ref)
    assert t_nat.log == t_ref.log
    del sfn


class RecordingTranscript(DeviceTranscript):
    """keeps the points / scalars and the challenges in the order the prover produced them"""

    def __init__(self, modulus):
        super().__init__(modulus)
        self.points, self.scalars, self.challenges = [], [], []

    def write_point(self, jac):
        self.points.append(np.ascontiguousarray(jac, dtype=np.uint64)[:8].copy())
        super().write_point(jac)

    def write_scalar(self, limbs):
        self.scalars.append(np.ascontiguousarray(limbs, dtype=np.uint64).copy())
        super().write_scalar(limbs)

    def squeeze_challenge_scalar(self):
        c = super().squeeze_challenge_scalar()
        self.challenges.append(c)
        return c


@pytest.mark.parametrize("curve,k", [("vesta", 4), ("pallas", 6)])
def test_ipa_proof_verifies(curve, k):
    """what the reference's own tests pin: the verifier accepts.  The opening produced on the GPU (single-call prover, device
    commitments) is checked with the oracle's restatement of commitment::verify_proof, which shares nothing with the prover
    restatement but the group law; a tampered evaluation must be rejected."""
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    rnd = random.Random(0xFEF1 + k)
    g_l = cpu_ref.gen_bases(curve, 17, 5, n, threads=4)
    w_l = cpu_ref.gen_bases(curve, 515151, 1, 1, threads=1)
    u_l = cpu_ref.gen_bases(curve, 626262, 1, 1, threads=1)
    params = poly.Params(curve, k, g_l, g_l, w_l, u=u_l)
    p_poly = [rnd.randrange(fs.m) for _ in range(n)]
    s_poly = [rnd.randrange(fs.m) for _ in range(n)]
    p_blind, s_blind, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
    draws = iter([rnd.randrange(fs.m) for _ in range(2 * k)])
    p_l = np.array([fs.limbs(v) for v in p_poly], np.uint64)
    commitment = params.commit(p_l, np.array(fs.limbs(p_blind), np.uint64))           # P, on the device
    tr = RecordingTranscript(fs.m)
    c, f = ipa.create_proof_native(params, lambda: next(draws), tr, to_dev(p_l), p_blind, x3, np.array([fs.limbs(v) for v in s_poly], np.uint64), s_blind)
    assert len(tr.points) == 1 + 2 * k and len(tr.challenges) == 2 + k
    pts = [cv.affine_from_limbs(p) for p in tr.points]
    xi, z, ch = tr.challenges[0], tr.challenges[1], tr.challenges[2:]
    rounds = [(pts[1 + 2 * j], pts[2 + 2 * j]) for j in range(k)]
    v = 0
    for cf in reversed(p_poly):
        v = (v * x3 + cf) % fs.m
    args = (cv, k, [cv.affine_from_limbs(r) for r in g_l], cv.affine_from_limbs(w_l[0]), cv.affine_from_limbs(u_l[0]), cv.affine_from_limbs(commitment[:8]))
    assert o.ipa_verify_proof(*args, x3, v, pts[0], xi, z, rounds, ch, c, f)
    assert not o.ipa_verify_proof(*args, x3, (v + 1) % fs.m, pts[0], xi, z, rounds, ch, c, f)
    assert not o.ipa_verify_proof(*args, x3, v, pts[0], xi, z, rounds, ch, c, (f + 1) % fs.m)


def test_ipa_base_set_forms_agree():
    """trh_ipa_create_proof over g || w (the MSMs convert g || w || u per proof), over a resident g || w || u WITHOUT tables and over
    one WITH fixed-base tables: the same transcript; a g || w || u set whose last point is not u is refused"""
    curve, k = "pallas", 7
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    rnd = random.Random(0x1FA5E7)
    g_l = cpu_ref.gen_bases(curve, 23, 11, n, threads=4)
    w_l = cpu_ref.gen_bases(curve, 515152, 1, 1, threads=1)
    u_l = cpu_ref.gen_bases(curve, 626263, 1, 1, threads=1)
    p_l, s_l = synth.field_elements(0x1FA1, n), synth.field_elements(0x1FA2, n)
    p_blind, s_blind, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
    draws = [rnd.randrange(fs.m) for _ in range(2 * k)]

    def run(params):
        it = iter(draws)
        tr = RecordingTranscript(fs.m)
        c, f = ipa.create_proof_native(params, lambda: next(it), tr, to_dev(p_l), p_blind, x3, s_l, s_blind)
        return c, f, [p.tolist() for p in tr.points], tr.challenges

    plain = poly.Params(curve, k, g_l, g_l, w_l, u=u_l, precompute=False)
    assert len(plain.ipa_bases()) == n + 1
    want = run(plain)
    gwu = np.concatenate([g_l, w_l, u_l])
    no_table = poly.Params(curve, k, g_l, g_l, w_l, u=u_l, precompute=False)
    no_table._ipa = api.Bases.from_host(curve, gwu)
    assert run(no_table) == want
    tables = poly.Params(curve, k, g_l, g_l, w_l, u=u_l, precompute=True)
    assert len(tables.ipa_bases()) == n + 2 and int(api.lib().trh_bases_precomputed_window_bits(tables.ipa_bases().handle)) > 0
    assert run(tables) == want
    wrong = poly.Params(curve, k, g_l, g_l, w_l, u=u_l, precompute=False)
    wrong._ipa = api.Bases.from_host(curve, np.concatenate([g_l, w_l, w_l]))
    with pytest.raises(api.TrhError):
        run(wrong)


@pytest.mark.parametrize("curve,k", [("vesta", 5)])
def test_multiopen_create_proof_vs_oracle(curve, k):
    """poly::multiopen::create_proof on resident polynomials (x1 folds per point set, kate_division, x2 fold, commitment, evaluations,
    x4 fold, IPA) against the oracle's restatement: identical transcripts"""
    from tiny_ram_halo2_amd import multiopen
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    rnd = random.Random(0x0BE1)
    g_l = cpu_ref.gen_bases(curve, 23, 9, n, threads=4)
    w_l = cpu_ref.gen_bases(curve, 818181, 1, 1, threads=1)
    u_l = cpu_ref.gen_bases(curve, 929292, 1, 1, threads=1)
    params = poly.Params(curve, k, g_l, g_l, w_l, u=u_l)
    keys = ["a", "b", "c", "z", "h"]
    polys = {key: [rnd.randrange(fs.m) for _ in range(n)] for key in keys}
    blinds = {key: rnd.randrange(fs.m) for key in keys}
    x = rnd.randrange(fs.m)
    wgen = fs.omega(k)
    xw, xwinv = x * wgen % fs.m, x * pow(wgen, -1, fs.m) % fs.m
    # advice at x; one column also at the next and previous rows; the product column at x and x omega; h at x: three point sets
    queries = [(x, "a"), (x, "b"), (xw, "b"), (xwinv, "b"), (x, "c"), (x, "z"), (xw, "z"), (x, "h"), (xw, "b")]
    draws = [rnd.randrange(fs.m) for _ in range(2 + n + 1 + 2 * k)]
    it1, it2 = iter(draws), iter(draws)
    t_dev, t_ref = DeviceTranscript(fs.m), OracleTranscript(cv)
    dev_polys = {key: to_dev(np.array([fs.limbs(v) for v in polys[key]], np.uint64)) for key in keys}
    got = multiopen.create_proof(params, lambda: next(it1), t_dev, queries, dev_polys, blinds)
    want = o.multiopen_create_proof(cv, k, [cv.affine_from_limbs(r) for r in g_l], cv.affine_from_limbs(w_l[0]), cv.affine_from_limbs(u_l[0]),
                                    lambda: next(it2), t_ref, queries, polys, blinds)
    assert got == want
    assert len(t_dev.log) == len(t_ref.log) == 1 + 3 + (1 + 2 * k + 2)   # q' commitment, three set evaluations, then the IPA
    assert t_dev.log == t_ref.log
