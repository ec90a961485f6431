"""Host-side mirror of `halo2_proofs::poly::{EvaluationDomain, commitment::Params}` for the
MSM / NTT call sites of `create_proof` (reference: /root/reference/src/test_utils.rs:21-49 reaches
them through keygen_vk / keygen_pk / create_proof; crate pinned at Cargo.lock:619-621).

Polynomials live in device memory as torch int64 tensors of shape (..., n, 4) holding the
u64 Montgomery limbs (torch is used for allocation, zero-padding and slicing only -- every field
operation is a libtrh kernel on the tensor's stream).  Same method names, argument meaning and
assertions as the Rust types:

    EvaluationDomain(field, j, k)          EvaluationDomain::new(j, k)
        .lagrange_to_coeff(a)              iFFT with omega^-1, then * n^-1
        .coeff_to_extended(a)              zeta-coset shift, zero-pad to 2^extended_k, FFT
        .extended_to_coeff(a)              iFFT, * 2^-extended_k, inverse coset shift, truncate
        .divide_by_vanishing_poly(a)       pointwise * (X^n - 1)^-1 on the coset (period 2^(ek-k))
    Params(curve, k, g, g_lagrange, w, u)  commitment::Params
        .commit(poly, r) / .commit_lagrange(poly, r)      MSM over (g | g_lagrange) ‖ w, n + 1 pairs
"""
from __future__ import annotations

import os

import numpy as np

from . import api

_MODULUS = {
    "fp": 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
    "fq": 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
}
_ROOT_OF_UNITY = {  # pasta_curves ROOT_OF_UNITY (2^32-th root), canonical
    "fp": 0x2BCE74DEAC30EBDA362120830561F81AEA322BF2B7BB7584BDAD6FABD87EA32F,
    "fq": 0x2DE6A9B8746D3F589E5C4DFD492AE26E9BB97EA3C106F049A70E2C1102B6D05F,
}
_ZETA = {  # pasta_curves ZETA (cube root of unity), canonical
    "fp": 0x12CCCA834ACDBA712CAAD5DC57AAB1B01D1F8BD237AD31491DAD5EBDFDFE4AB9,
    "fq": 0x06819A58283E528E511DB4D81CF70F5A0FED467D47C033AF2AA9D2E050AA0E4F,
}
S = 32


def _mont(field: str, x: int) -> np.ndarray:
    m = _MODULUS[field]
    v = x % m * ((1 << 256) % m) % m
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _stream(t):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


class EvaluationDomain:
    def __init__(self, field: str, j: int, k: int):
        m = _MODULUS[field]
        self.field, self.k, self.n = field, k, 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        ext_omega = _ROOT_OF_UNITY[field]
        for _ in range(ek, S):
            ext_omega = ext_omega * ext_omega % m
        omega = ext_omega
        for _ in range(k, ek):
            omega = omega * omega % m
        self.omega, self.omega_inv = omega, pow(omega, -1, m)
        self.extended_omega, self.extended_omega_inv = ext_omega, pow(ext_omega, -1, m)
        self.g_coset = _ZETA[field]
        self.g_coset_inv = self.g_coset * self.g_coset % m
        self.ifft_divisor = pow(2, -k, m)
        self.extended_ifft_divisor = pow(2, -ek, m)
        # t(X) = X^n - 1 on the coset zeta * extended_omega^i: period 2^(ek - k), inverted
        orig = pow(self.g_coset, self.n, m)
        step = pow(ext_omega, self.n, m)
        t_eval, cur = [], orig
        while True:
            t_eval.append(cur)
            cur = cur * step % m
            if cur == orig:
                break
        assert len(t_eval) == 1 << (ek - k)
        self.t_evaluations = [pow((t - 1) % m, -1, m) for t in t_eval]
        # limb forms
        self._w = {name: _mont(field, getattr(self, name)) for name in ("omega", "omega_inv", "extended_omega", "extended_omega_inv",
                                                                        "ifft_divisor", "extended_ifft_divisor")}
        self._into_coset = np.stack([_mont(field, 1), _mont(field, self.g_coset), _mont(field, self.g_coset_inv)])
        self._from_coset = np.stack([_mont(field, 1), _mont(field, self.g_coset_inv), _mont(field, self.g_coset)])
        self._t_inv = np.stack([_mont(field, t) for t in self.t_evaluations])

    def extended_len(self) -> int:
        return 1 << self.extended_k

    @staticmethod
    def _batch(a):
        assert a.shape[-1] == 4
        b = 1
        for d in a.shape[:-2]:
            b *= d
        return b

    # The transforms run through libtrh's C++ EvaluationDomain (csrc/domain.hip), which derives the same
    # constants from ROOT_OF_UNITY / ZETA; the big-int values above are kept for inspection and are
    # cross-checked against the library in handle().
    def handle(self):
        if getattr(self, "_h", None) is None:
            import ctypes
            h = ctypes.c_void_p()
            api._check(api.lib().trh_domain_create(api.FIELD_ID[self.field], self.quotient_poly_degree + 1, self.k, ctypes.byref(h)))
            assert api.lib().trh_domain_extended_k(h) == self.extended_k
            out = np.zeros(4, dtype=np.uint64)
            for which, name in enumerate(("omega", "omega_inv", "extended_omega", "extended_omega_inv", "ifft_divisor", "extended_ifft_divisor")):
                api._check(api.lib().trh_domain_constant(h, which, api._p(out)))
                assert (out == self._w[name]).all(), name
            self._h = h
        return self._h

    def reserve(self, batch: int):
        """trh_domain_reserve: tables and scratch of the per-column transforms for batches of `batch` polynomials (setup time)"""
        api._check(api.lib().trh_domain_reserve(self.handle(), batch))

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                api.lib().trh_domain_destroy(self._h)
        except Exception:
            pass

    def lagrange_to_coeff(self, a):
        """in place on a (..., n, 4) device tensor; returns a"""
        assert a.shape[-2] == self.n
        api._check(api.lib().trh_domain_lagrange_to_coeff(self.handle(), api._devptr(a), self._batch(a), _stream(a)))
        return a

    def coeff_to_lagrange(self, a):
        assert a.shape[-2] == self.n
        api.ntt_dev(self.field, a, self.k, self._w["omega"], batch=self._batch(a), stream=_stream(a))
        return a

    def coeff_to_extended(self, a, out=None):
        """(..., n, 4) coefficients -> (..., 2^extended_k, 4) tensor of coset evaluations (a new one unless `out` is given)"""
        import torch
        assert a.shape[-2] == self.n
        a = a.contiguous()
        shape = a.shape[:-2] + (self.extended_len(), 4)
        if out is not None:
            assert out.is_contiguous() and out.numel() >= int(np.prod(shape))
            ext = out.reshape(-1)[: int(np.prod(shape))].reshape(shape)
        else:
            ext = torch.empty(shape, dtype=a.dtype, device=a.device)
        api._check(api.lib().trh_domain_coeff_to_extended(self.handle(), api._devptr(a), api._devptr(ext), self._batch(a), _stream(a)))
        return ext

    def extended_to_coeff(self, a):
        """in place iFFT + inverse coset shift on (..., 2^extended_k, 4); returns the truncated view
        of length n * quotient_poly_degree"""
        assert a.shape[-2] == self.extended_len()
        api._check(api.lib().trh_domain_extended_to_coeff(self.handle(), api._devptr(a), self._batch(a), _stream(a)))
        return a[..., : self.n * self.quotient_poly_degree, :]

    def divide_by_vanishing_poly(self, a):
        assert a.shape[-2] == self.extended_len()
        api._check(api.lib().trh_domain_divide_by_vanishing_poly(self.handle(), api._devptr(a), self._batch(a), _stream(a)))
        return a


    # ---- the extended domain as coset blocks (csrc/domain.hip): entry q of block r is entry q * 2^(extended_k - k) + r of
    # coeff_to_extended's output; the quotient h(X) needs quotient_poly_degree blocks of every column ----
    def coeff_to_extended_blocks(self, a, n_blocks: int | None = None, out=None):
        """(..., n, 4) coefficients -> (..., n_blocks, n, 4) coset evaluations (default: the quotient_poly_degree blocks h(X) needs)"""
        import torch
        nb = self.quotient_poly_degree if n_blocks is None else n_blocks
        assert a.shape[-2] == self.n and 1 <= nb <= (1 << (self.extended_k - self.k))
        a = a.contiguous()
        shape = a.shape[:-2] + (nb, self.n, 4)
        if out is not None:
            assert out.is_contiguous() and out.numel() >= int(np.prod(shape))
            ext = out.reshape(-1)[: int(np.prod(shape))].reshape(shape)
        else:
            ext = torch.empty(shape, dtype=a.dtype, device=a.device)
        api._check(api.lib().trh_domain_coeff_to_extended_blocks(self.handle(), api._devptr(a), api._devptr(ext), self._batch(a), nb, _stream(a)))
        return ext

    def blocks_to_quotient(self, num_blocks, divide_by_vanishing: bool = True):
        """(quotient_poly_degree, n, 4) values of the quotient's numerator on blocks 0 .. (overwritten) -> (quotient_poly_degree * n, 4)
        coefficients of h(X): extended_to_coeff(divide_by_vanishing_poly(.)) for a numerator the vanishing polynomial divides"""
        import torch
        d = self.quotient_poly_degree
        assert num_blocks.is_contiguous() and num_blocks.numel() == d * self.n * 4
        out = torch.empty((d * self.n, 4), dtype=num_blocks.dtype, device=num_blocks.device)
        api._check(api.lib().trh_domain_blocks_to_quotient(self.handle(), api._devptr(num_blocks), api._devptr(out), 1 if divide_by_vanishing else 0, _stream(num_blocks)))
        return out

    # ---- the same operations on HOST polynomials (numpy (n, 4) uint64 arrays, as the Rust host holds its `Polynomial`s):
    # csrc/hostio.hip pipelines the columns over PCIe ----
    def lagrange_to_coeff_host(self, columns):
        """in place on a list of host columns"""
        for a in columns:
            assert a.dtype == np.uint64 and a.flags.c_contiguous and a.shape == (self.n, 4)
        api._check(api.lib().trh_domain_lagrange_to_coeff_host(self.handle(), api._ptr_array(columns), len(columns)))
        return columns

    def coeff_to_extended_host(self, coeffs, out=None):
        """list of (n, 4) host coefficient arrays -> list of (2^extended_k, 4) host arrays"""
        for a in coeffs:
            assert a.dtype == np.uint64 and a.flags.c_contiguous and a.shape == (self.n, 4)
        ext = out if out is not None else [np.empty((self.extended_len(), 4), dtype=np.uint64) for _ in coeffs]
        api._check(api.lib().trh_domain_coeff_to_extended_host(self.handle(), api._ptr_array(coeffs), api._ptr_array(ext), len(coeffs)))
        return ext

    def coeff_to_extended_blocks_host(self, coeffs, n_blocks: int | None = None, out=None):
        """list of (n, 4) host coefficient arrays -> list of (n_blocks, n, 4) host arrays (the coset-block layout)"""
        nb = self.quotient_poly_degree if n_blocks is None else n_blocks
        for a in coeffs:
            assert a.dtype == np.uint64 and a.flags.c_contiguous and a.shape == (self.n, 4)
        ext = out if out is not None else [np.empty((nb, self.n, 4), dtype=np.uint64) for _ in coeffs]
        api._check(api.lib().trh_domain_coeff_to_extended_blocks_host(self.handle(), api._ptr_array(coeffs), api._ptr_array(ext), len(coeffs), nb))
        return ext

    def blocks_to_quotient_host(self, num_blocks, divide_by_vanishing: bool = True):
        """(quotient_poly_degree, n, 4) host values of the numerator on blocks 0 .. -> (quotient_poly_degree * n, 4) coefficients of h(X)"""
        d = self.quotient_poly_degree
        assert num_blocks.dtype == np.uint64 and num_blocks.flags.c_contiguous and num_blocks.size == d * self.n * 4
        out = np.empty((d * self.n, 4), dtype=np.uint64)
        api._check(api.lib().trh_domain_blocks_to_quotient_host(self.handle(), api._p(num_blocks), api._p(out), 1 if divide_by_vanishing else 0))
        return out

    def extended_to_coeff_host(self, a, divide_by_vanishing_first: bool = False):
        """in place on one (2^extended_k, 4) host array; returns the truncated view of n * quotient_poly_degree coefficients"""
        assert a.dtype == np.uint64 and a.flags.c_contiguous and a.shape == (self.extended_len(), 4)
        api._check(api.lib().trh_domain_extended_to_coeff_host(self.handle(), api._p(a), 1 if divide_by_vanishing_first else 0))
        return a[: self.n * self.quotient_poly_degree]


class Params:
    """commitment::Params with device-resident bases: g ‖ w and g_lagrange ‖ w (n + 1 points each)."""

    def __init__(self, curve: str, k: int, g, g_lagrange, w, u=None, precompute: bool = True):
        """precompute: attach libtrh's fixed-base tables (trh_bases_precompute) to both base sets -- Params are
        fixed for the life of a proving key, and create_proof commits ~500 columns against them."""
        self.curve, self.k, self.n = curve, k, 1 << k
        g = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, 8)
        gl = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(-1, 8)
        w = np.ascontiguousarray(w, dtype=np.uint64).reshape(1, 8)
        assert g.shape[0] == self.n and gl.shape[0] == self.n
        self.w, self.u = w, u
        self._g = api.Bases.from_host(curve, np.concatenate([g, w]))
        self._g_lagrange = api.Bases.from_host(curve, np.concatenate([gl, w]))
        if precompute:
            self.precompute()

    def precompute(self):
        """fixed-base tables for commit / commit_lagrange (full-range MSMs of n + 1 pairs); a no-op when the set is
        outside the supported range (W (n + 1) <= 2^24, i.e. k <= 19)"""
        for b in (self._g, self._g_lagrange):
            try:
                b.precompute(0)
            except api.TrhError:
                pass
        self._ipa = None
        self.ipa_bases()

    def reserve(self, batch: int):
        """trh_bases_reserve over the resident sets: the calling context's MSM scratch for commit batches of `batch` columns and the
        opening's round MSMs (setup time: the first batch of a process then allocates nothing)"""
        for b in (self._g, self._g_lagrange):
            api._check(api.lib().trh_bases_reserve(b.handle, self.n + 1, batch))
        ipa = self.ipa_bases()
        if ipa is not self._g:
            api._check(api.lib().trh_bases_reserve(ipa.handle, self.n + 2, 2))

    def ipa_bases(self):
        """g || w || u as ONE resident set with fixed-base tables: `trh_ipa_create_proof` then runs every MSM of the opening in
        fixed-base mode (csrc/ipa.hip).  Built on first use (a download of g || w, an upload, the table kernel: once per Params);
        falls back to the g || w set when u is unknown or the tables do not fit."""
        cached = getattr(self, "_ipa", None)
        if cached is not None:
            return cached
        self._ipa = self._g
        u = getattr(self, "u", None)
        if u is not None and int(api.lib().trh_bases_precomputed_window_bits(self._g.handle)) != 0:  # only for Params that use tables at all
            gwu = np.concatenate([self._g.download(), np.ascontiguousarray(u, dtype=np.uint64).reshape(1, 8)])
            b = api.Bases.from_host(self.curve, gwu)
            try:
                b.precompute(int(os.environ.get("TRH_IPA_TABLE_BITS", "0")))
                self._ipa = b
            except api.TrhError:
                pass
        return self._ipa

    @staticmethod
    def g_lagrange_from_g(curve: str, k: int, g_dev):
        """The heavy step of `Params::new(k)`: g_lagrange = batch_normalize(n^-1 * best_fft(g, omega^-1, k)) with
        best_fft over curve points.  g_dev: device tensor (n, 8) of affine generators; returns a new tensor.
        (The hash-to-curve derivation of g itself stays on the host.)"""
        sf = api.SCALAR_FIELD[curve]
        m = _MODULUS[sf]
        omega = _ROOT_OF_UNITY[sf]
        for _ in range(k, S):
            omega = omega * omega % m
        out = g_dev.clone()
        api.point_fft_dev(curve, out, k, _mont(sf, pow(omega, -1, m)), scale=_mont(sf, pow(1 << k, -1, m)), stream=_stream(out))
        return out

    def _commit(self, bases, poly, r):
        import torch
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        if isinstance(poly, np.ndarray):
            sc = np.concatenate([poly.astype(np.uint64).reshape(-1, 4), r.reshape(1, 4)])
            assert sc.shape[0] == self.n + 1
            return bases.msm(sc)
        assert poly.shape[-2] == self.n
        sc = torch.cat([poly.reshape(self.n, 4), torch.from_numpy(r.view(np.int64)).to(poly.device).reshape(1, 4)])
        return bases.msm_dev(sc, self.n + 1, stream=_stream(sc))

    def commit(self, poly, r):
        """poly in coefficient form (host numpy (n, 4) or device tensor), r the blind: MSM of n + 1 pairs"""
        return self._commit(self._g, poly, r)

    def commit_lagrange(self, poly, r):
        return self._commit(self._g_lagrange, poly, r)

    def commit_lagrange_batch(self, polys, blinds):
        """polys: device tensor (batch, n, 4); blinds: (batch, 4) host limbs -> (batch, 12) points"""
        batch = polys.shape[0]
        assert polys.shape[1] == self.n and polys.is_contiguous()
        return self._g_lagrange.commit_batch_dev(polys, self.n, batch, np.ascontiguousarray(blinds, dtype=np.uint64).reshape(batch, 4), stream=_stream(polys))

    def commit_lagrange_batch_host(self, polys, blinds):
        """polys: list of host (n, 4) columns; blinds: (batch, 4) -> (batch, 12) points (trh_commit_batch_host)"""
        return self._g_lagrange.commit_batch_host(polys, blinds)

    def commit_batch_host(self, polys, blinds):
        return self._g.commit_batch_host(polys, blinds)

    def commit_batch(self, polys, blinds):
        """coefficient-form counterpart of commit_lagrange_batch"""
        batch = polys.shape[0]
        assert polys.shape[1] == self.n and polys.is_contiguous()
        return self._g.commit_batch_dev(polys, self.n, batch, np.ascontiguousarray(blinds, dtype=np.uint64).reshape(batch, 4), stream=_stream(polys))
