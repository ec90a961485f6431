"""ORACLE (test infrastructure, NOT product code): ctypes binding of oracle/libtrh_oracle.so
(the C++ restatement in oracle/cpu_ref.cpp).  Arrays are numpy uint64, little-endian limbs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtrh_oracle.so")
_lib = None

FIELD_ID = {"fp": 0, "fq": 1}
CURVE_ID = {"pallas": 0, "vesta": 1}
OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "neg": 4, "inv": 5, "to_mont": 6, "from_mont": 7}

_u64p = ctypes.POINTER(ctypes.c_uint64)
_POINT_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, _u64p)
_SCALAR_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, _u64p)
_SQUEEZE_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, _u64p)


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE, "libtrh_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_field_op.argtypes = [ctypes.c_int, ctypes.c_int, _u64p, _u64p, _u64p, ctypes.c_size_t]
        _lib.orc_point_op.argtypes = [ctypes.c_int, ctypes.c_int, _u64p, _u64p, _u64p]
        _lib.orc_to_affine.argtypes = [ctypes.c_int, _u64p, _u64p]
        _lib.orc_scalar_mul.argtypes = [ctypes.c_int, _u64p, _u64p, _u64p]
        _lib.orc_best_multiexp.argtypes = [ctypes.c_int, _u64p, _u64p, ctypes.c_size_t, ctypes.c_int, _u64p]
        _lib.orc_best_fft.argtypes = [ctypes.c_int, _u64p, _u64p, ctypes.c_uint32, ctypes.c_int]
        _lib.orc_gen_bases.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int, _u64p]
        _lib.orc_hardware_threads.restype = ctypes.c_int
        _lib.orc_gen_bases_hashed.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int, _u64p]
        _lib.orc_hashed_scalars.argtypes = [ctypes.c_uint64, ctypes.c_size_t, _u64p]
        _lib.orc_best_fft_points.argtypes = [ctypes.c_int, _u64p, _u64p, ctypes.c_uint32, ctypes.c_int]
        _lib.orc_scale_points.argtypes = [ctypes.c_int, _u64p, _u64p, ctypes.c_size_t, ctypes.c_int, _u64p]
        _lib.orc_scale_points_each.argtypes = [ctypes.c_int, _u64p, _u64p, ctypes.c_size_t, ctypes.c_int, _u64p]
        _lib.orc_ipa_create_proof.argtypes = [ctypes.c_int, ctypes.c_uint32, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p, ctypes.c_void_p,
                                              _POINT_CB, _SCALAR_CB, _SQUEEZE_CB, _SQUEEZE_CB, ctypes.c_void_p, ctypes.c_int, _u64p, _u64p]
        _lib.orc_eval_polynomial.argtypes = [ctypes.c_int, _u64p, ctypes.c_size_t, _u64p, _u64p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_u64p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def hardware_threads() -> int:
    return int(lib().orc_hardware_threads())


def field_op(field: str, op: str, a, b=None):
    a = _c(a).reshape(-1, 4)
    b = None if b is None else _c(b).reshape(-1, 4)
    out = np.empty_like(a)
    lib().orc_field_op(FIELD_ID[field], OPS[op], _p(a), _p(b), _p(out), a.shape[0])
    return out


def point_op(curve: str, op: str, p, q=None):
    p = _c(p).reshape(12)
    q = None if q is None else _c(q).reshape(-1)
    out = np.empty(12, dtype=np.uint64)
    lib().orc_point_op(CURVE_ID[curve], {"add": 0, "madd": 1, "dbl": 2}[op], _p(p), _p(q), _p(out))
    return out


def to_affine(curve: str, xyz):
    xyz = _c(xyz).reshape(12)
    out = np.empty(8, dtype=np.uint64)
    lib().orc_to_affine(CURVE_ID[curve], _p(xyz), _p(out))
    return out


def scalar_mul(curve: str, base_xy, k_canonical):
    out = np.empty(12, dtype=np.uint64)
    b, k = _c(base_xy).reshape(8), _c(k_canonical).reshape(4)
    lib().orc_scalar_mul(CURVE_ID[curve], _p(b), _p(k), _p(out))
    return out


def best_multiexp(curve: str, coeffs_mont, bases_xy, threads: int = 1):
    """coeffs: n x 4 Montgomery limbs (scalar field), bases: n x 8.  Returns Jacobian 12 limbs."""
    c, b = _c(coeffs_mont).reshape(-1, 4), _c(bases_xy).reshape(-1, 8)
    assert c.shape[0] == b.shape[0]  # reference: assert_eq!(coeffs.len(), bases.len())
    out = np.empty(12, dtype=np.uint64)
    lib().orc_best_multiexp(CURVE_ID[curve], _p(c), _p(b), c.shape[0], threads, _p(out))
    return out


def best_fft(field: str, a, omega, log_n: int, threads: int = 1):
    """Returns a new array (the C routine works in place on a copy)."""
    a = _c(a).reshape(-1, 4).copy()
    assert a.shape[0] == 1 << log_n  # reference: assert_eq!(a.len(), 1 << log_n)
    w = _c(omega).reshape(4)
    lib().orc_best_fft(FIELD_ID[field], _p(a), _p(w), log_n, threads)
    return a


def gen_bases(curve: str, s0: int, d: int, n: int, threads: int = 0):
    out = np.empty((n, 8), dtype=np.uint64)
    lib().orc_gen_bases(CURVE_ID[curve], s0, d, n, threads or hardware_threads(), _p(out))
    return out


def gen_bases_hashed(curve: str, seed: int, n: int, threads: int = 0):
    """n bases with no arithmetic structure between them: P_i = h_i * G, h_i a hash of (seed, i)"""
    out = np.empty((n, 8), dtype=np.uint64)
    lib().orc_gen_bases_hashed(CURVE_ID[curve], seed, n, threads or hardware_threads(), _p(out))
    return out


def hashed_scalars(seed: int, n: int):
    """the discrete logs h_i of gen_bases_hashed (canonical limbs)"""
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_hashed_scalars(seed, n, _p(out))
    return out


def best_fft_points(curve: str, points_xy, omega, log_n: int, threads: int = 0):
    """halo2_proofs::arithmetic::best_fft::<C::Curve> on affine PODs; returns the transformed copy, normalised to affine"""
    a = _c(points_xy).reshape(-1, 8).copy()
    assert a.shape[0] == 1 << log_n
    w = _c(omega).reshape(4)
    lib().orc_best_fft_points(CURVE_ID[curve], _p(a), _p(w), log_n, threads or hardware_threads())
    return a


def scale_points(curve: str, base_xy, scalars_mont, threads: int = 0):
    """out[i] = scalars[i] * base, affine"""
    s = _c(scalars_mont).reshape(-1, 4)
    b = _c(base_xy).reshape(8)
    out = np.empty((s.shape[0], 8), dtype=np.uint64)
    lib().orc_scale_points(CURVE_ID[curve], _p(b), _p(s), s.shape[0], threads or hardware_threads(), _p(out))
    return out


def scale_points_each(curve: str, bases_xy, scalars_mont, threads: int = 0):
    """out[i] = scalars[i] * bases[i], affine"""
    s = _c(scalars_mont).reshape(-1, 4)
    b = _c(bases_xy).reshape(-1, 8)
    assert s.shape[0] == b.shape[0]
    out = np.empty((s.shape[0], 8), dtype=np.uint64)
    lib().orc_scale_points_each(CURVE_ID[curve], _p(b), _p(s), s.shape[0], threads or hardware_threads(), _p(out))
    return out


def eval_polynomial(field: str, poly, x):
    p, xx = _c(poly).reshape(-1, 4), _c(x).reshape(4)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_eval_polynomial(FIELD_ID[field], _p(p), p.shape[0], _p(xx), _p(out))
    return out


def ipa_create_proof(curve: str, k: int, g_xy, w_xy, u_xy, rng, transcript, p_poly, p_blind, x3, s_poly, s_blind, threads: int = 0):
    """poly::commitment::prover::create_proof restated in C++ (literal: G' is collapsed with scalar multiplications).
    Everything is Montgomery limbs; `transcript` has write_point(xyz[12]) / write_scalar(limbs[4]) / squeeze_challenge_scalar() -> limbs,
    `rng()` -> limbs.  Returns (c, f) as limb arrays."""
    n = 1 << k
    g, w, u = _c(g_xy).reshape(n, 8), _c(w_xy).reshape(8), _c(u_xy).reshape(8)
    pp, sp = _c(p_poly).reshape(n, 4), _c(s_poly).reshape(n, 4)
    pb, xx, sb = _c(p_blind).reshape(4), _c(x3).reshape(4), _c(s_blind).reshape(4)
    out_c, out_f = np.empty(4, np.uint64), np.empty(4, np.uint64)

    def _wp(_ctx, ptr):
        transcript.write_point(np.array([ptr[i] for i in range(12)], dtype=np.uint64))

    def _ws(_ctx, ptr):
        transcript.write_scalar(np.array([ptr[i] for i in range(4)], dtype=np.uint64))

    def _sq(_ctx, ptr):
        v = transcript.squeeze_challenge_scalar()
        for i in range(4):
            ptr[i] = int(v[i])

    def _rng(_ctx, ptr):
        v = rng()
        for i in range(4):
            ptr[i] = int(v[i])

    cbs = (_POINT_CB(_wp), _SCALAR_CB(_ws), _SQUEEZE_CB(_sq), _SQUEEZE_CB(_rng))
    rc = lib().orc_ipa_create_proof(CURVE_ID[curve], k, _p(g), _p(w), _p(u), _p(pp), _p(pb), _p(xx), _p(sp), _p(sb), None, cbs[0], cbs[1], cbs[2], cbs[3], None,
                                    threads or hardware_threads(), _p(out_c), _p(out_f))
    if rc != 0:
        raise ZeroDivisionError("ipa_create_proof: zero round challenge (u_j.invert().unwrap())")
    return out_c, out_f


class EvaluationDomain:
    """halo2_proofs 0.2.0 poly::EvaluationDomain (src/poly/domain.rs) on (n, 4) Montgomery limb arrays: the constants come from
    oracle/pasta.py::EvaluationDomain, the transforms are the C++ best_fft and element-wise field ops -- the same steps as the
    big-int restatement (pinned against it in tests/test_oracle.py), fast enough for k = 18 / extended_k = 21."""

    def __init__(self, field: str, j: int, k: int, threads: int = 0):
        import pasta
        self.field, self.f = field, pasta.FIELDS[field]
        self.c = pasta.EvaluationDomain(self.f, j, k)
        self.k, self.n, self.extended_k = k, 1 << k, self.c.extended_k
        self.threads = threads or hardware_threads()

    def _lim(self, v):
        return np.array(self.f.limbs(v), dtype=np.uint64)

    def _scale_periodic(self, a, factors):
        fac = np.array([self.f.limbs(v) for v in factors], dtype=np.uint64)
        reps = (a.shape[0] + len(factors) - 1) // len(factors)
        return field_op(self.field, "mul", a, np.tile(fac, (reps, 1))[: a.shape[0]])

    def lagrange_to_coeff(self, a):
        a = _c(a).reshape(self.n, 4)
        return self._scale_periodic(best_fft(self.field, a, self._lim(self.c.omega_inv), self.k, self.threads), [self.c.ifft_divisor])

    def coeff_to_extended(self, a):
        a = _c(a).reshape(self.n, 4)
        ext = np.zeros((1 << self.extended_k, 4), dtype=np.uint64)
        ext[: self.n] = self._scale_periodic(a, [1, self.c.g_coset, self.c.g_coset_inv])
        return best_fft(self.field, ext, self._lim(self.c.extended_omega), self.extended_k, self.threads)

    def divide_by_vanishing_poly(self, a):
        return self._scale_periodic(_c(a).reshape(1 << self.extended_k, 4), self.c.t_evaluations)

    def extended_to_coeff(self, a):
        a = _c(a).reshape(1 << self.extended_k, 4)
        a = self._scale_periodic(best_fft(self.field, a, self._lim(self.c.extended_omega_inv), self.extended_k, self.threads), [self.c.extended_ifft_divisor])
        a = self._scale_periodic(a, [1, self.c.g_coset_inv, self.c.g_coset])
        return a[: self.n * self.c.quotient_poly_degree]


def prefix_product(field: str, a):
    """out[i] = prod_{j < i} a[j], out[0] = 1 (the running product of the permutation / lookup z columns), limb arrays.
    Sequential by definition; done in Python ints (a few hundred thousand mulmods per second)."""
    import pasta
    f = pasta.FIELDS[field]
    a = _c(a).reshape(-1, 4)
    r, vals = 1, []
    for row in a:
        vals.append(r)
        r = r * f.from_limbs(row) % f.m
    return np.array([f.limbs(v) for v in vals], dtype=np.uint64)
