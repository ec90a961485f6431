"""GPU parity tests (-m gpu) for the EvaluationDomain / Params mirrors:
device results vs the big-int restatement in oracle/pasta.py, plus the algebraic properties the
domain offers (coset round trip; division by the vanishing polynomial inverts multiplication)."""
import os
import random

import numpy as np
import pytest
import torch

import cpu_ref
import pasta as o
from tiny_ram_halo2_amd import api, poly, synth

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


def limbs(f, vals):
    return np.array([f.limbs(v) for v in vals], dtype=np.uint64)


@pytest.mark.parametrize("field", ["fp", "fq"])
@pytest.mark.parametrize("k,j", [(4, 4), (6, 7), (8, 3)])
def test_domain_vs_oracle(field, k, j):
    f = o.FIELDS[field]
    rng = random.Random(k * 100 + j)
    dom, ref = poly.EvaluationDomain(field, j, k), o.EvaluationDomain(f, j, k)
    assert dom.extended_k == ref.extended_k and dom.omega == ref.omega and dom.t_evaluations == ref.t_evaluations
    batch = 3
    vals = [[rng.randrange(f.m) for _ in range(dom.n)] for _ in range(batch)]
    a = to_dev(np.stack([limbs(f, v) for v in vals]))
    # lagrange_to_coeff
    coeff = dom.lagrange_to_coeff(a.clone())
    want = [ref.lagrange_to_coeff(v) for v in vals]
    got = to_host(coeff)
    for b in range(batch):
        assert [f.from_limbs(r) for r in got[b]] == want[b]
    # coeff_to_extended
    ext = dom.coeff_to_extended(coeff)
    want_ext = [ref.coeff_to_extended(w) for w in want]
    got = to_host(ext)
    for b in range(batch):
        assert [f.from_limbs(r) for r in got[b]] == want_ext[b]
    # divide_by_vanishing_poly
    div = dom.divide_by_vanishing_poly(ext.clone())
    got = to_host(div)
    want_div = [ref.divide_by_vanishing_poly(w) for w in want_ext]
    for b in range(batch):
        assert [f.from_limbs(r) for r in got[b]] == want_div[b]
    # extended_to_coeff (of the undivided evaluations: must give back the coefficients, zero-padded)
    back = dom.extended_to_coeff(ext.clone())
    got = to_host(back)
    for b in range(batch):
        w = ref.extended_to_coeff(want_ext[b])
        assert [f.from_limbs(r) for r in got[b]] == w
        assert w[: dom.n] == want[b] and all(v == 0 for v in w[dom.n:])


def test_domain_roundtrip_k14():
    """size-independent property at a create_proof-like size: extended_to_coeff(coeff_to_extended(a)) == a ‖ 0"""
    field, k, j = "fp", 14, 7
    dom = poly.EvaluationDomain(field, j, k)
    a = synth.field_elements(0xD0, 2 * dom.n).reshape(2, dom.n, 4)
    d = to_dev(a)
    ext = dom.coeff_to_extended(d)
    assert ext.shape == (2, dom.extended_len(), 4)
    back = to_host(dom.extended_to_coeff(ext))
    assert (back[:, : dom.n] == a).all() and (back[:, dom.n:] == 0).all()
    # lagrange_to_coeff inverts coeff_to_lagrange
    lag = dom.coeff_to_lagrange(d.clone())
    assert (to_host(dom.lagrange_to_coeff(lag)) == a).all()


@pytest.mark.parametrize("curve", ["vesta", "pallas"])
def test_params_commit(curve):
    k = 8
    n = 1 << k
    sf = api.SCALAR_FIELD[curve]
    g = cpu_ref.gen_bases(curve, 11, 3, n, threads=4)
    gl = cpu_ref.gen_bases(curve, 1234567, 5, n, threads=4)
    w = cpu_ref.gen_bases(curve, 999, 1, 1, threads=1)
    params = poly.Params(curve, k, g, gl, w)
    p = synth.field_elements(0xC0, n)
    r = synth.field_elements(0xC1, 1)[0]
    want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, np.concatenate([p, r[None]]), np.concatenate([g, w]), threads=4))
    assert (params.commit(p, r)[:8] == want).all()
    assert (params.commit(to_dev(p), r)[:8] == want).all()
    want_l = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, np.concatenate([p, r[None]]), np.concatenate([gl, w]), threads=4))
    assert (params.commit_lagrange(p, r)[:8] == want_l).all()
    # batch of columns
    batch = 4
    ps = synth.field_elements(0xC2, n * batch).reshape(batch, n, 4)
    rs = synth.field_elements(0xC3, batch)
    got = params.commit_lagrange_batch(to_dev(ps), rs)
    for i in range(batch):
        wl = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, np.concatenate([ps[i], rs[i][None]]), np.concatenate([gl, w]), threads=4))
        assert (got[i, :8] == wl).all()
    got_c = params.commit_batch(to_dev(ps), rs)  # coefficient-form counterpart, also without the fixed-base tables
    plain = poly.Params(curve, k, g, gl, w, precompute=False)
    assert (plain.commit_batch(to_dev(ps), rs) == got_c).all() and (plain.commit_lagrange_batch(to_dev(ps), rs) == got).all()
    for i in range(batch):
        wc = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, np.concatenate([ps[i], rs[i][None]]), np.concatenate([g, w]), threads=4))
        assert (got_c[i, :8] == wc).all()
    # commit(a) == commit_lagrange(lagrange form of a) when g_lagrange is the Lagrange basis of g is a
    # property of Params::new (a "next" row); here the two base sets are independent.
    del sf


@pytest.mark.parametrize("curve,k", [("vesta", 1), ("vesta", 4), ("pallas", 6)])
def test_params_g_lagrange_point_fft(curve, k):
    """Params::new's curve-point FFT: device g_lagrange vs the big-int restatement, and the
    defining property commit(coeffs) == commit_lagrange(evaluations) for a random polynomial."""
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    g_l = cpu_ref.gen_bases(curve, 0xBEEF + k, 11, n, threads=2)
    g_l[n - 1] = 0 if k >= 4 else g_l[n - 1]          # an identity generator must survive the transform
    got = to_host(poly.Params.g_lagrange_from_g(curve, k, to_dev(g_l)))
    want = o.params_g_lagrange(cv, [cv.affine_from_limbs(r) for r in g_l], k)
    for i in range(n):
        assert cv.affine_from_limbs(got[i]) == want[i], i
    # commit(a) with g == commit_lagrange(a evaluated on the domain) with g_lagrange (blind 0)
    rnd = random.Random(k)
    coeffs = [rnd.randrange(fs.m) for _ in range(n)]
    evals = o.best_fft(fs, coeffs, fs.omega(k), k)
    w = cpu_ref.gen_bases(curve, 5, 1, 1, threads=1)
    params = poly.Params(curve, k, g_l, got, w)
    zero = np.zeros(4, np.uint64)
    c1 = params.commit(limbs(fs, coeffs), zero)
    c2 = params.commit_lagrange(limbs(fs, evals), zero)
    assert (c1 == c2).all()


@pytest.mark.parametrize("field", ["fp", "fq"])
@pytest.mark.parametrize("n", [1, 5, 64, 65, 4096, 4097, 100003])
def test_batch_invert_and_prefix_product(field, n):
    """grand-product primitives of the permutation / lookup arguments: ff::BatchInvert semantics
    (zeros stay zero) and z[i] = prod_{j<i} a[j], vs the CPU restatement / big-int oracle"""
    f = o.FIELDS[field]
    a = synth.field_elements(0x5CA0 + n, n)
    if n >= 5:
        a[3] = 0
        a[n - 1] = 0
    d = to_dev(a)
    api.batch_invert_dev(field, d, n)
    got = to_host(d)
    nz = (a != 0).any(axis=1)
    want = np.zeros_like(a)
    want[nz] = cpu_ref.field_op(field, "inv", a[nz])
    assert (got == want).all()
    # inverse * original == 1 where non-zero (independent of the oracle's inversion)
    one = np.array(f.limbs(1), np.uint64)
    assert (cpu_ref.field_op(field, "mul", got[nz], a[nz]) == one).all()

    b = synth.field_elements(0x5CB0 + n, n)
    db, dout = to_dev(b), torch.empty((n, 4), dtype=torch.int64, device="cuda")
    api.prefix_product_dev(field, db, dout, n)
    z = to_host(dout)
    run = 1
    vals = [f.from_limbs(r) for r in b]
    for i in range(n):
        if i < 200 or i % 997 == 0 or i == n - 1:
            assert f.from_limbs(z[i]) == run, i
        run = run * vals[i] % f.m
    # z[i+1] == z[i] * b[i] everywhere (size-independent property)
    if n > 1:
        assert (cpu_ref.field_op(field, "mul", z[:-1], b[:-1]) == z[1:]).all()


@pytest.mark.parametrize("field", ["fp", "fq"])
@pytest.mark.parametrize("k,j", [(11, 2), (11, 3), (12, 5), (12, 6), (11, 17), (13, 9)])
def test_domain_fused_passes_vs_unfused_primitives(field, k, j):
    """From log_n = 11 the pointwise steps are fused into the NTT passes (zero-padding + coset shift on the loads of
    pass 0 with the all-zero stages skipped, x 2^-k / inverse coset shift on the final store).  The same results must come
    out of the separate primitives (scale kernels + plain NTT, themselves checked against the oracle) and of the CPU
    restatement of best_fft."""
    import cpu_ref
    f = o.FIELDS[field]
    dom = poly.EvaluationDomain(field, j, k)
    n, N, batch = dom.n, dom.extended_len(), 2
    a = synth.field_elements(0xF05E + 31 * k + j, batch * n).reshape(batch, n, 4)
    # lagrange_to_coeff
    got = to_host(dom.lagrange_to_coeff(to_dev(a)))
    ref = to_dev(a)
    api.ntt_dev(field, ref, k, dom._w["omega_inv"], batch=batch)
    api.field_scale_dev(field, ref, batch * n, dom._w["ifft_divisor"])
    coeff = to_host(ref)
    assert (got == coeff).all()
    assert (got[0] == cpu_ref.field_op(field, "mul", cpu_ref.best_fft(field, a[0], dom._w["omega_inv"], k, threads=4),
                                       np.tile(dom._w["ifft_divisor"], (n, 1)))).all()
    # coeff_to_extended
    ext = dom.coeff_to_extended(to_dev(coeff))
    padded = np.zeros((batch, N, 4), np.uint64)
    padded[:, :n] = coeff
    ref = to_dev(padded)
    api.field_scale_rows_dev(field, ref, batch, N, n, dom._into_coset)
    api.ntt_dev(field, ref, dom.extended_k, dom._w["extended_omega"], batch=batch)
    want_ext = to_host(ref)
    assert (to_host(ext) == want_ext).all()
    shifted = cpu_ref.field_op(field, "mul", coeff[1], dom._into_coset[np.arange(n) % 3])
    padded1 = np.zeros((N, 4), np.uint64)
    padded1[:n] = shifted
    assert (want_ext[1] == cpu_ref.best_fft(field, padded1, dom._w["extended_omega"], dom.extended_k, threads=4)).all()
    # extended_to_coeff
    back = to_host(dom.extended_to_coeff(ext.clone()))
    assert (back[:, :n] == coeff).all() and (back[:, n:] == 0).all()
    ref = to_dev(want_ext)
    api.ntt_dev(field, ref, dom.extended_k, dom._w["extended_omega_inv"], batch=batch)
    api.field_scale_dev(field, ref, batch * N, dom._w["extended_ifft_divisor"])
    api.field_scale_rows_dev(field, ref, batch, N, N, dom._from_coset)
    assert (back == to_host(ref)[:, : back.shape[1]]).all()  # the Rust code truncates to n * quotient_poly_degree coefficients


@pytest.mark.parametrize("field", ["fp", "fq"])
def test_poly_eval_batch(field):
    """arithmetic::eval_polynomial (Horner) for a batch of device polynomials; at x = omega^i it must agree with the NTT"""
    f = o.FIELDS[field]
    rng = random.Random(0xE7A1)
    for n, batch in ((1, 2), (5, 3), (257, 4), (3000, 2)):
        polys = [[rng.randrange(f.m) for _ in range(n)] for _ in range(batch)]
        x = rng.randrange(f.m)
        d = to_dev(np.stack([limbs(f, p) for p in polys]))
        got = api.poly_eval_batch_dev(field, d, n, batch, np.array(f.limbs(x), np.uint64))
        for b in range(batch):
            acc = 0
            for cf in reversed(polys[b]):
                acc = (acc * x + cf) % f.m
            assert f.from_limbs(got[b]) == acc, (n, b)
    # x = 0 and x = 1
    d = to_dev(limbs(f, [7, 8, 9])[None])
    assert f.from_limbs(api.poly_eval_batch_dev(field, d, 3, 1, np.array(f.limbs(0), np.uint64))[0]) == 7
    assert f.from_limbs(api.poly_eval_batch_dev(field, d, 3, 1, np.array(f.limbs(1), np.uint64))[0]) == 24
    # size-independent property at 2^16: evaluations at omega^i are entries of the forward transform
    k, batch = 16, 3
    n = 1 << k
    a = synth.field_elements(0xE7A2, batch * n).reshape(batch, n, 4)
    w = f.omega(k)
    fwd = to_dev(a)
    api.ntt_dev(field, fwd, k, np.array(f.limbs(w), np.uint64), batch=batch)
    fwd = to_host(fwd)
    for i in (0, 1, 12345, n - 1):
        got = api.poly_eval_batch_dev(field, to_dev(a), n, batch, np.array(f.limbs(pow(w, i, f.m)), np.uint64))
        assert (got == fwd[:, i]).all(), i


@pytest.mark.parametrize("field", ["fp", "fq"])
def test_multiopen_lincomb_and_kate_division(field):
    """poly/multiopen/prover.rs building blocks: sum_b x1^b p_b and arithmetic::kate_division (synthetic division by X - z)"""
    from tiny_ram_halo2_amd import multiopen
    f = o.FIELDS[field]
    rng = random.Random(0x0FE7)
    for n, batch in ((2, 1), (7, 3), (300, 5), (4097, 2), (70000, 2)):
        polys = [[rng.randrange(f.m) for _ in range(n)] for _ in range(batch)]
        if n > 4:
            polys[0][n - 1] = 0; polys[0][2] = f.m - 1
        x1 = rng.randrange(f.m)
        coeffs = [pow(x1, b, f.m) for b in range(batch)]
        d = to_dev(np.stack([limbs(f, p) for p in polys]))
        comb = multiopen.lincomb(field, d, coeffs)
        want = [sum(coeffs[b] * polys[b][i] for b in range(batch)) % f.m for i in range(n)]
        got = [f.from_limbs(r) for r in to_host(comb)]
        assert got == want
        for z in (rng.randrange(f.m), 1, f.m - 1, 0):
            q = [f.from_limbs(r) for r in to_host(multiopen.KateDivider(field, n, z, comb.device).divide(comb))]
            # the Rust loop: q[n-2] = a[n-1]; q[i-1] = a[i] + z q[i]
            ref, tmp = [0] * (n - 1), 0
            for i in range(n - 1, 0, -1):
                tmp = (want[i] + z * tmp) % f.m
                ref[i - 1] = tmp
            assert q == ref, (n, z)
    # prefix sums on their own
    a = [rng.randrange(f.m) for _ in range(5000)]
    d, out = to_dev(limbs(f, a)), to_dev(limbs(f, [0] * 5000))
    api._check(api.lib().trh_field_prefix_sum_dev(api.FIELD_ID[field], api._devptr(d), api._devptr(out), 5000, None))
    acc, want = 0, []
    for v in a:
        want.append(acc)
        acc = (acc + v) % f.m
    assert [f.from_limbs(r) for r in to_host(out)] == want


@pytest.mark.parametrize("field,k,j", [("fp", 5, 6), ("fq", 9, 4), ("fp", 11, 6), ("fq", 12, 6), ("fp", 13, 3), ("fp", 18, 6)])
def test_extended_domain_as_coset_blocks(field, k, j):
    """trh_domain_coeff_to_extended_blocks / trh_domain_blocks_to_quotient against EvaluationDomain's own functions:
    (1) block r, entry q == coeff_to_extended(a)[q 2^(extended_k - k) + r] -- all 2^(extended_k - k) blocks and the quotient's j - 1,
        fused (k >= 11, incl. the reference's k = 18 / extended_k = 21) and unfused (small k) forms, batch of ragged size;
    (2) for a numerator the vanishing polynomial divides -- N = h (X^n - 1) with a random h of (j - 1) n coefficients, evaluated on the
        extended coset by the oracle -- blocks_to_quotient of its first j - 1 blocks == extended_to_coeff(divide_by_vanishing_poly(N))
        == h, coefficient for coefficient."""
    import cpu_ref
    f = o.FIELDS[field]
    dom = poly.EvaluationDomain(field, j, k)
    cd = cpu_ref.EvaluationDomain(field, j, k)
    n, ek = 1 << k, dom.extended_k
    N, step, D = 1 << ek, 1 << (ek - k), j - 1
    batch = 3
    a = synth.field_elements(0xB10C + k, batch * n).reshape(batch, n, 4)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    want = [np.asarray(cd.coeff_to_extended(a[i])).reshape(N, 4) for i in (0, batch - 1)]
    for nb in (step, D):
        got = dom.coeff_to_extended_blocks(d_a, nb)
        torch.cuda.synchronize()
        g = got.cpu().numpy().view(np.uint64).reshape(batch, nb, n, 4)
        for wi, i in enumerate((0, batch - 1)):
            for r in range(nb):
                assert (g[i, r] == want[wi][r::step]).all(), (nb, i, r)
    # (2) quotient from j - 1 blocks
    h = synth.field_elements(0x40 + k, D * n)
    lim = lambda v: np.array(f.limbs(v), np.uint64)  # noqa: E731
    # h on the extended coset: zero-pad to 2^extended_k, zeta shift, best_fft (EvaluationDomain::coeff_to_extended on a longer polynomial)
    padded = np.zeros((N, 4), dtype=np.uint64)
    padded[: D * n] = h
    zs = [1, cd.c.g_coset, cd.c.g_coset_inv]
    fac = np.array([f.limbs(zs[i % 3]) for i in range(3)], dtype=np.uint64)
    padded[: D * n] = cpu_ref.field_op(field, "mul", padded[: D * n], np.tile(fac, (D * n // 3 + 1, 1))[: D * n])
    h_ext = cpu_ref.best_fft(field, padded, lim(cd.c.extended_omega), ek, cpu_ref.hardware_threads())
    # numerator = h * t with t = X^n - 1 on the coset (period 2^(extended_k - k)): multiply by the inverses' inverses
    t_vals = np.array([f.limbs(pow(v, -1, f.m)) for v in cd.c.t_evaluations], dtype=np.uint64)
    num = cpu_ref.field_op(field, "mul", h_ext, np.tile(t_vals, (N // len(t_vals), 1)))
    ref = np.asarray(cd.extended_to_coeff(cd.divide_by_vanishing_poly(num.copy()))).reshape(-1, 4)
    assert (ref[: D * n] == h).all()                        # the oracle's own chain returns h
    blocks = np.stack([num[r::step] for r in range(D)])     # the numerator on blocks 0 .. j - 2
    d_num = torch.from_numpy(np.ascontiguousarray(blocks).view(np.int64)).cuda()
    got_h = dom.blocks_to_quotient(d_num, divide_by_vanishing=True)
    torch.cuda.synchronize()
    assert (got_h.cpu().numpy().view(np.uint64) == h).all()
    # without the division: h's own values in, h out
    d_hv = torch.from_numpy(np.ascontiguousarray(np.stack([h_ext[r::step] for r in range(D)])).view(np.int64)).cuda()
    assert (dom.blocks_to_quotient(d_hv, divide_by_vanishing=False).cpu().numpy().view(np.uint64) == h).all()
