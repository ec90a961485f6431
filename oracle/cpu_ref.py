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
