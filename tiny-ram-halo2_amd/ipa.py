"""Host-side mirror of `halo2_proofs::poly::commitment::prover::create_proof` (the IPA opening that
`poly::multiopen::create_proof` ends with; reference call site /root/reference/src/test_utils.rs:41-49,
SURVEY.md section 8 row a7).  Vectors p', b and the generators G' stay in device memory for the k
rounds; per round only the two points L_j, R_j and the challenge cross the host boundary.

The transcript (BLAKE2b, on the Rust host) and the prover's randomness are injected:
    transcript.write_point(jacobian12), .write_scalar(limbs4), .squeeze_challenge_scalar() -> int (canonical)
    rng() -> int (canonical scalar)
so the flow can be replayed bit-for-bit against the oracle restatement.
"""
from __future__ import annotations

import numpy as np

from . import api
from .poly import _MODULUS, _mont, _stream


def _canon(field: str, limbs) -> int:
    m = _MODULUS[field]
    v = 0
    for i, w in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(4)):
        v |= int(w) << (64 * i)
    return v * pow(1 << 256, -1, m) % m


def create_proof_native(params, rng, transcript, p_poly, p_blind: int, x3: int, s_poly, s_blind: int):
    """The same opening through the single C entry point `trh_ipa_create_proof` (csrc/ipa.hip): the round loop
    and all host-side scalar arithmetic run in C++, the transcript and randomness are callbacks."""
    import ctypes

    import torch

    curve, k, n = params.curve, params.k, params.n
    sf = api.SCALAR_FIELD[curve]

    def limbs_of(ptr, count):
        return np.array([ptr[i] for i in range(count)], dtype=np.uint64)

    def put(ptr, limbs):
        for i in range(4):
            ptr[i] = int(limbs[i])

    wp = api.WRITE_POINT_FN(lambda ctx, p: transcript.write_point(limbs_of(p, 12)))
    ws = api.WRITE_SCALAR_FN(lambda ctx, p: transcript.write_scalar(limbs_of(p, 4)))
    sq = api.SQUEEZE_FN(lambda ctx, out: put(out, _mont(sf, transcript.squeeze_challenge_scalar())))
    rn = api.RNG_FN(lambda ctx, out: put(out, _mont(sf, rng())))
    tr = api.Transcript(None, wp, ws, sq)
    # (np.require copies only when s_poly is not already contiguous and writable: an unconditional .copy() of the 8 MiB was 1 ms of every k = 18 opening)
    s_dev = torch.from_numpy(np.require(s_poly, dtype=np.uint64, requirements=["C", "W"]).view(np.int64)).to(p_poly.device)
    out_c, out_f = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    u = np.ascontiguousarray(params.u, dtype=np.uint64).reshape(8)
    bases = params.ipa_bases() if hasattr(params, "ipa_bases") else params._g
    api._check(api.lib().trh_ipa_create_proof(bases.handle, api._p(u), k, api._devptr(p_poly), api._p(_mont(sf, p_blind)), api._p(_mont(sf, x3)),
                                              api._devptr(s_dev), api._p(_mont(sf, s_blind)), ctypes.byref(tr), rn, None, _stream(p_poly),
                                              api._p(out_c), api._p(out_f)))
    return _canon(sf, out_c), _canon(sf, out_f)


def create_proof(params, rng, transcript, p_poly, p_blind: int, x3: int, s_poly=None, s_blind: int | None = None):
    """p_poly: device tensor (n, 4) of coefficients (Montgomery limbs); p_blind, x3 canonical ints.
    s_poly (host numpy (n, 4), random with s(x3) = 0 enforced here) and s_blind default to rng draws."""
    import torch

    curve, k, n = params.curve, params.k, params.n
    sf = api.SCALAR_FIELD[curve]
    m = _MODULUS[sf]
    dev = p_poly.device
    st = _stream(p_poly)

    def dev_of(host_limbs):
        return torch.from_numpy(np.ascontiguousarray(host_limbs, dtype=np.uint64).view(np.int64).copy()).to(dev)

    # b = (1, x3, x3^2, ...)
    b = torch.empty((n, 4), dtype=torch.int64, device=dev)
    api.powers_dev(sf, b, n, _mont(sf, x3), stream=st)

    # s(X): random with a root at x3
    if s_poly is None:
        s_poly = np.stack([_mont(sf, rng()) for _ in range(n)])
    s_dev = dev_of(s_poly)
    s_at_x3 = _canon(sf, api.inner_product_dev(sf, s_dev, b, n, stream=st))
    c0 = (_canon(sf, s_poly[0]) - s_at_x3) % m
    s_dev[0] = dev_of(_mont(sf, c0))
    if s_blind is None:
        s_blind = rng()
    s_commitment = params.commit(s_dev, _mont(sf, s_blind))
    transcript.write_point(s_commitment)
    xi = transcript.squeeze_challenge_scalar()
    z = transcript.squeeze_challenge_scalar()

    # p'(X) = p(X) + xi s(X) - v,  v = p'(x3) before the subtraction
    p_prime = p_poly.clone()
    api.axpy_dev(sf, p_prime, s_dev, n, _mont(sf, xi), stream=st)
    v = _canon(sf, api.inner_product_dev(sf, p_prime, b, n, stream=st))
    p0 = (_canon(sf, p_prime[0].cpu().numpy().view(np.uint64)) - v) % m
    p_prime[0] = dev_of(_mont(sf, p0))
    f = (s_blind * xi + p_blind) % m

    # G' starts as a private copy of params.g (n points); u and w are the last two bases of the 2-term MSMs
    g_prime = torch.empty((n, 8), dtype=torch.int64, device=dev)
    g_host = params._g.download(0, n)
    g_prime.copy_(dev_of(g_host))
    uw = api.Bases.from_host(curve, np.concatenate([np.asarray(params.u, dtype=np.uint64).reshape(1, 8), params.w]))
    gp = api.Bases.wrap_device(curve, g_prime, n)

    for j in range(k):
        half = 1 << (k - j - 1)
        l_j = gp.msm_dev(p_prime[half:2 * half], half, offset=0, stream=st)          # <p'[half..], G'[..half]>
        r_j = gp.msm_dev(p_prime[:half], half, offset=half, stream=st)               # <p'[..half], G'[half..]>
        value_l = _canon(sf, api.inner_product_dev(sf, p_prime[half:2 * half], b[:half], half, stream=st))
        value_r = _canon(sf, api.inner_product_dev(sf, p_prime[:half], b[half:2 * half], half, stream=st))
        l_rand, r_rand = rng(), rng()
        l_j = api.point_sum(curve, np.stack([l_j, uw.msm(np.stack([_mont(sf, value_l * z % m), _mont(sf, l_rand)]))]))
        r_j = api.point_sum(curve, np.stack([r_j, uw.msm(np.stack([_mont(sf, value_r * z % m), _mont(sf, r_rand)]))]))
        transcript.write_point(l_j)
        transcript.write_point(r_j)
        u_j = transcript.squeeze_challenge_scalar()
        u_inv = pow(u_j, -1, m)
        # collapse p', b and the generators
        api.axpy_dev(sf, p_prime[:half], p_prime[half:2 * half], half, _mont(sf, u_inv), stream=st)
        api.axpy_dev(sf, b[:half], b[half:2 * half], half, _mont(sf, u_j), stream=st)
        api.bases_fold_dev(curve, g_prime[:half], g_prime[half:2 * half], half, _mont(sf, u_j), stream=st)
        f = (f + l_rand * u_inv + r_rand * u_j) % m

    c = _canon(sf, p_prime[0].cpu().numpy().view(np.uint64))
    transcript.write_scalar(_mont(sf, c))
    transcript.write_scalar(_mont(sf, f))
    return c, f
