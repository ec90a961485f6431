"""ctypes binding of libtrh.so plus a host-side mirror of the halo2_proofs interface for the
MSM / NTT / commitment hot path (same names and argument meaning as the Rust functions the
reference reaches through create_proof, /root/reference/src/test_utils.rs:21-49):

    best_multiexp(coeffs, bases)            halo2_proofs::arithmetic::best_multiexp
    best_fft(a, omega, log_n)               halo2_proofs::arithmetic::best_fft
    Params(curve, g, g_lagrange, w, u)      halo2_proofs::poly::commitment::Params
        .commit(poly, r) / .commit_lagrange(poly, r)
    EvaluationDomain(field, k, j)           halo2_proofs::poly::EvaluationDomain
        .lagrange_to_coeff / .coeff_to_extended / .extended_to_coeff

Arrays are numpy uint64 (little-endian limbs, Montgomery form) on the host, or raw device
pointers (ints / torch tensors' data_ptr()) for the *_dev forms.  The library has no CPU
fallback: loading fails loudly if libtrh.so is missing, and every compute call raises
TrhError when no MI355X is bound.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRH_LIB_PATH") or os.path.join(_HERE, "libtrh.so")  # the override serves same-box A/B measurements of two builds

PALLAS, VESTA = 0, 1
FP, FQ = 0, 1
CURVE_ID = {"pallas": PALLAS, "vesta": VESTA}
FIELD_ID = {"fp": FP, "fq": FQ}
# scalar field of each curve (pallas: base Fp / scalar Fq; vesta: base Fq / scalar Fp)
SCALAR_FIELD = {"pallas": "fq", "vesta": "fp"}
BASE_FIELD = {"pallas": "fp", "vesta": "fq"}
FIELD_OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "neg": 4, "inv": 5, "to_mont": 6, "from_mont": 7}
POINT_OPS = {"add": 0, "madd": 1, "dbl": 2, "q4_add": 3, "q4_dbl": 4}  # q4_*: the quad-lane arithmetic of csrc/curve_q4.h

_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p


class TrhError(RuntimeError):
    pass


WRITE_POINT_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64))
WRITE_SCALAR_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64))
SQUEEZE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64))
RNG_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64))


class Transcript(ctypes.Structure):
    _fields_ = [("ctx", ctypes.c_void_p), ("write_point", WRITE_POINT_FN), ("write_scalar", WRITE_SCALAR_FN),
                ("squeeze_challenge_scalar", SQUEEZE_FN)]


class Timing(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_float), ("digits_ms", ctypes.c_float), ("sort_ms", ctypes.c_float),
                ("accumulate_ms", ctypes.c_float), ("reduce_ms", ctypes.c_float),
                ("window_bits", ctypes.c_int), ("windows", ctypes.c_int), ("accumulate_kernel_ms", ctypes.c_float)]


class IoStats(ctypes.Structure):
    _fields_ = [("h2d_bytes", ctypes.c_double), ("d2h_bytes", ctypes.c_double), ("h2d_seconds", ctypes.c_double), ("d2h_seconds", ctypes.c_double),
                ("h2d_zero_bytes", ctypes.c_double)]


_SIGNATURES = {
    "trh_init": ([ctypes.c_int], ctypes.c_int),
    "trh_shutdown": ([], None),
    "trh_last_error": ([], ctypes.c_char_p),
    "trh_device_count": ([], ctypes.c_int),
    "trh_version": ([], ctypes.c_char_p),
    "trh_set_option": ([ctypes.c_char_p, ctypes.c_char_p], ctypes.c_int),
    "trh_get_option": ([ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)], ctypes.c_int),
    "trh_init_multi": ([ctypes.POINTER(ctypes.c_int), ctypes.c_int], ctypes.c_int),
    "trh_group_size": ([], ctypes.c_int),
    "trh_group_peer_access": ([], ctypes.c_int),
    "trh_set_shard_min": ([ctypes.c_size_t], ctypes.c_int),
    "trh_ctx_create": ([ctypes.c_int, ctypes.POINTER(_vp)], ctypes.c_int),
    "trh_ctx_destroy": ([_vp], None),
    "trh_ctx_set_current": ([_vp], ctypes.c_int),
    "trh_ctx_device": ([_vp], ctypes.c_int),
    "trh_ctx_stream": ([_vp], _vp),
    "trh_best_multiexp_pallas": ([_u64p, _u64p, ctypes.c_size_t, _u64p], ctypes.c_int),
    "trh_best_multiexp_vesta": ([_u64p, _u64p, ctypes.c_size_t, _u64p], ctypes.c_int),
    "trh_best_fft_fp": ([_u64p, _u64p, ctypes.c_uint32], ctypes.c_int),
    "trh_best_fft_fq": ([_u64p, _u64p, ctypes.c_uint32], ctypes.c_int),
    "trh_bases_create_pallas": ([_u64p, ctypes.c_size_t, ctypes.POINTER(_vp)], ctypes.c_int),
    "trh_bases_create_vesta": ([_u64p, ctypes.c_size_t, ctypes.POINTER(_vp)], ctypes.c_int),
    "trh_bases_wrap_device": ([ctypes.c_int, _vp, ctypes.c_size_t, ctypes.POINTER(_vp)], ctypes.c_int),
    "trh_bases_generate": ([ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t, ctypes.POINTER(_vp)], ctypes.c_int),
    "trh_bases_download": ([_vp, ctypes.c_size_t, ctypes.c_size_t, _u64p], ctypes.c_int),
    "trh_bases_device_ptr": ([_vp], _vp),
    "trh_bases_len": ([_vp], ctypes.c_size_t),
    "trh_bases_shards": ([_vp], ctypes.c_int),
    "trh_bases_destroy": ([_vp], None),
    "trh_bases_precompute": ([_vp, ctypes.c_int], ctypes.c_int),
    "trh_bases_precomputed_window_bits": ([_vp], ctypes.c_int),
    "trh_bases_reserve": ([_vp, ctypes.c_size_t, ctypes.c_size_t], ctypes.c_int),
    "trh_msm": ([_vp, ctypes.c_size_t, _u64p, ctypes.c_size_t, ctypes.c_int, _u64p], ctypes.c_int),
    "trh_msm_dev": ([_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.c_int, _vp, _u64p], ctypes.c_int),
    "trh_msm_dev_enqueue": ([_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.c_int, _vp], ctypes.c_int),
    "trh_msm_dev_finish": ([_vp, _vp, _u64p], ctypes.c_int),
    "trh_msm_batch_dev": ([_vp, ctypes.c_size_t, _vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, _vp, _u64p], ctypes.c_int),
    "trh_commit_batch_dev": ([_vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _u64p, _vp, _u64p], ctypes.c_int),
    "trh_msm_set_window_bits": ([ctypes.c_int], ctypes.c_int),
    "trh_point_sum": ([ctypes.c_int, _u64p, ctypes.c_size_t, _u64p], ctypes.c_int),
    "trh_ntt_dev": ([ctypes.c_int, _vp, ctypes.c_uint32, _u64p, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_field_scale_dev": ([ctypes.c_int, _vp, ctypes.c_size_t, _u64p, _vp], ctypes.c_int),
    "trh_field_scale_periodic_dev": ([ctypes.c_int, _vp, ctypes.c_size_t, _u64p, ctypes.c_uint32, _vp], ctypes.c_int),
    "trh_field_scale_rows_dev": ([ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, _u64p, ctypes.c_uint32, _vp], ctypes.c_int),
    "trh_field_inner_product_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, _vp, _u64p], ctypes.c_int),
    "trh_field_axpy_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, _u64p, _vp], ctypes.c_int),
    "trh_field_powers_dev": ([ctypes.c_int, _vp, ctypes.c_size_t, _u64p, _vp], ctypes.c_int),
    "trh_bases_fold_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, _u64p, _vp], ctypes.c_int),
    "trh_point_fft_dev": ([ctypes.c_int, _vp, ctypes.c_uint32, _u64p, _u64p, _vp], ctypes.c_int),
    "trh_domain_create": ([ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_vp)], ctypes.c_int),
    "trh_domain_destroy": ([_vp], None),
    "trh_domain_reserve": ([_vp, ctypes.c_size_t], ctypes.c_int),
    "trh_domain_extended_k": ([_vp], ctypes.c_uint32),
    "trh_domain_constant": ([_vp, ctypes.c_int, _u64p], ctypes.c_int),
    "trh_domain_lagrange_to_coeff": ([_vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_domain_coeff_to_extended": ([_vp, _vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_domain_extended_to_coeff": ([_vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_domain_divide_by_vanishing_poly": ([_vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_ipa_create_proof": ([_vp, _u64p, ctypes.c_uint32, _vp, _u64p, _u64p, _vp, _u64p, ctypes.POINTER(Transcript), RNG_FN, _vp, _vp, _u64p, _u64p], ctypes.c_int),
    "trh_poly_eval_batch_dev": ([ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_size_t, _u64p, _vp, _u64p], ctypes.c_int),
    "trh_field_batch_invert_dev": ([ctypes.c_int, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_field_batch_invert_mul_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_product_terms_dev": ([ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.c_size_t, _vp, _vp], ctypes.c_int),
    "trh_field_prefix_product_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_lookup_permute_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, _vp, _vp, _vp], ctypes.c_int),
    "trh_lookup_permute_batch_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, _vp, _vp, _vp], ctypes.c_int),
    "trh_expr_create": ([ctypes.c_int, _vp, ctypes.c_size_t, _u64p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                         ctypes.POINTER(ctypes.c_void_p)], ctypes.c_int),
    "trh_expr_destroy": ([_vp], None),
    "trh_expr_lds_slots": ([_vp], ctypes.c_uint32),
    "trh_expr_set_const": ([_vp, ctypes.c_uint32, _u64p], ctypes.c_int),
    "trh_expr_eval_dev": ([_vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_uint32, _vp], ctypes.c_int),
    "trh_field_prefix_product_rows_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_field_prefix_sum_dev": ([ctypes.c_int, _vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_poly_lincomb_dev": ([ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_size_t, _u64p, _vp, _vp], ctypes.c_int),
    "trh_poly_kate_division_dev": ([ctypes.c_int, _vp, ctypes.c_size_t, _vp, _vp, _vp, _vp, _vp], ctypes.c_int),
    "trh_stat": ([ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int),
    "trh_event_create": ([ctypes.POINTER(_vp)], ctypes.c_int),
    "trh_event_record": ([_vp, _vp], ctypes.c_int),
    "trh_event_elapsed_ms": ([_vp, _vp, ctypes.POINTER(ctypes.c_float)], ctypes.c_int),
    "trh_event_destroy": ([_vp], None),
    "trh_field_op_dev": ([ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_point_op_dev": ([ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, ctypes.c_size_t, _vp], ctypes.c_int),
    "trh_malloc": ([ctypes.POINTER(_vp), ctypes.c_size_t], ctypes.c_int),
    "trh_free": ([_vp], ctypes.c_int),
    "trh_pool_trim": ([], ctypes.c_int),
    "trh_pool_idle_bytes": ([], ctypes.c_size_t),
    "trh_memcpy_h2d": ([_vp, _vp, ctypes.c_size_t], ctypes.c_int),
    "trh_memcpy_d2h": ([_vp, _vp, ctypes.c_size_t], ctypes.c_int),
    "trh_stream_synchronize": ([_vp], ctypes.c_int),
    "trh_best_fft_batch_fp": ([ctypes.POINTER(_u64p), ctypes.c_size_t, _u64p, ctypes.c_uint32], ctypes.c_int),
    "trh_best_fft_batch_fq": ([ctypes.POINTER(_u64p), ctypes.c_size_t, _u64p, ctypes.c_uint32], ctypes.c_int),
    "trh_commit_batch_host": ([_vp, ctypes.POINTER(_u64p), ctypes.c_size_t, ctypes.c_size_t, _u64p, _u64p], ctypes.c_int),
    "trh_domain_quotient_blocks": ([_vp], ctypes.c_uint32),
    "trh_domain_coeff_to_extended_blocks": ([_vp, _vp, _vp, ctypes.c_size_t, ctypes.c_uint32, _vp], ctypes.c_int),
    "trh_domain_blocks_to_quotient": ([_vp, _vp, _vp, ctypes.c_int, _vp], ctypes.c_int),
    "trh_expr_eval_blocks_dev": ([_vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_uint32, _vp], ctypes.c_int),
    "trh_domain_lagrange_to_coeff_host": ([_vp, ctypes.POINTER(_u64p), ctypes.c_size_t], ctypes.c_int),
    "trh_domain_coeff_to_extended_host": ([_vp, ctypes.POINTER(_u64p), ctypes.POINTER(_u64p), ctypes.c_size_t], ctypes.c_int),
    "trh_domain_extended_to_coeff_host": ([_vp, _u64p, ctypes.c_int], ctypes.c_int),
    "trh_domain_coeff_to_extended_blocks_host": ([_vp, ctypes.POINTER(_u64p), ctypes.POINTER(_u64p), ctypes.c_size_t, ctypes.c_uint32], ctypes.c_int),
    "trh_domain_blocks_to_quotient_host": ([_vp, _u64p, _u64p, ctypes.c_int], ctypes.c_int),
    "trh_host_register": ([_vp, ctypes.c_size_t], ctypes.c_int),
    "trh_host_unregister": ([_vp], ctypes.c_int),
    "trh_host_alloc": ([ctypes.POINTER(_vp), ctypes.c_size_t], ctypes.c_int),
    "trh_host_free": ([_vp], ctypes.c_int),
    "trh_io_stats": ([ctypes.POINTER(IoStats), ctypes.c_int], ctypes.c_int),
    "trh_set_timing": ([ctypes.c_int], ctypes.c_int),
    "trh_last_timing": ([ctypes.POINTER(Timing)], ctypes.c_int),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Loads libtrh.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TrhError(f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback for the MSM/NTT path")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            try:
                fn = getattr(_lib, name)
            except AttributeError:
                if os.environ.get("TRH_LIB_PATH"):  # an older build loaded for a same-box A/B: entries added since are simply absent
                    continue
                raise
            fn.argtypes = argtypes
            fn.restype = restype
    return _lib


def _check(rc: int):
    if rc != 0:
        raise TrhError(f"libtrh error {rc}: {lib().trh_last_error().decode()}")


def init(device: int = 0):
    _check(lib().trh_init(device))


def set_option(name: str, value) -> None:
    """trh_set_option: a library switch; only while no context exists (before init / after shutdown)"""
    _check(lib().trh_set_option(name.encode(), str(int(value)).encode()))


def get_option(name: str) -> int:
    v = ctypes.c_long(0)
    _check(lib().trh_get_option(name.encode(), ctypes.byref(v)))
    return int(v.value)


def stat(name: str) -> int:
    v = ctypes.c_uint64(0)
    _check(lib().trh_stat(name.encode(), ctypes.byref(v)))
    return int(v.value)


def init_multi(devices):
    """trh_init_multi: the device group range-sharded base sets run on; devices[0] is the process default."""
    arr = (ctypes.c_int * len(devices))(*devices)
    _check(lib().trh_init_multi(arr, len(devices)))


def group_size() -> int:
    return int(lib().trh_group_size())


def set_shard_min(n_points: int):
    _check(lib().trh_set_shard_min(n_points))


class Context:
    """An independent libtrh context (own scratch, own lock) on one device; `with ctx:` binds the calling thread to it."""

    def __init__(self, device: int = 0):
        self.handle = _vp()
        _check(lib().trh_ctx_create(device, ctypes.byref(self.handle)))

    def bind(self):
        _check(lib().trh_ctx_set_current(self.handle))

    @staticmethod
    def unbind():
        _check(lib().trh_ctx_set_current(None))

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, *exc):
        self.unbind()

    def destroy(self):
        if self.handle:
            lib().trh_ctx_destroy(self.handle)
            self.handle = _vp()


def shutdown():
    lib().trh_shutdown()


def _c(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a if cols is None else a.reshape(-1, cols)


def _p(a):
    return a.ctypes.data_as(_u64p)


# ---------------------------------------------------------------------------------------
# halo2_proofs::arithmetic
# ---------------------------------------------------------------------------------------
def best_multiexp(curve: str, coeffs, bases) -> np.ndarray:
    """coeffs: (n, 4) Montgomery scalars, bases: (n, 8) affine.  Returns Jacobian (12,) with Z = 1."""
    c, b = _c(coeffs, 4), _c(bases, 8)
    assert c.shape[0] == b.shape[0]  # reference: assert_eq!(coeffs.len(), bases.len())
    out = np.zeros(12, dtype=np.uint64)
    fn = lib().trh_best_multiexp_pallas if curve == "pallas" else lib().trh_best_multiexp_vesta
    _check(fn(_p(c), _p(b), c.shape[0], _p(out)))
    return out


def best_fft(field: str, a, omega, log_n: int) -> np.ndarray:
    """In-place semantics of the Rust function; returns the transformed copy."""
    a = _c(a, 4).copy()
    assert a.shape[0] == 1 << log_n  # reference: assert_eq!(a.len(), 1 << log_n)
    w = _c(omega).reshape(4)
    fn = lib().trh_best_fft_fp if field == "fp" else lib().trh_best_fft_fq
    _check(fn(_p(a), _p(w), log_n))
    return a


def _ptr_array(arrays):
    """host columns -> C array of pointers (the arrays must stay alive for the call)"""
    return (_u64p * len(arrays))(*[_p(a) for a in arrays])


def best_fft_batch(field: str, columns, omega, log_n: int):
    """best_fft over a list of host columns (each (2^log_n, 4) uint64, C-contiguous), IN PLACE, pipelined over PCIe"""
    for a in columns:
        assert a.dtype == np.uint64 and a.flags.c_contiguous and a.size == 4 << log_n  # reference: assert_eq!(a.len(), 1 << log_n)
    w = _c(omega).reshape(4)
    fn = lib().trh_best_fft_batch_fp if field == "fp" else lib().trh_best_fft_batch_fq
    _check(fn(_ptr_array(columns), len(columns), _p(w), log_n))


def io_stats(reset: bool = False) -> dict:
    """bytes moved by the host-pointer entry points and the host seconds spent in the copies, per direction"""
    st = IoStats()
    _check(lib().trh_io_stats(ctypes.byref(st), 1 if reset else 0))
    return {k: getattr(st, k) for k, _ in IoStats._fields_}


def best_fft_inplace(field: str, a: np.ndarray, omega, log_n: int) -> np.ndarray:
    """the Rust signature: transforms the caller's (2^log_n, 4) uint64 array in place"""
    assert a.dtype == np.uint64 and a.flags.c_contiguous and a.size == 4 << log_n  # reference: assert_eq!(a.len(), 1 << log_n)
    w = _c(omega).reshape(4)
    fn = lib().trh_best_fft_fp if field == "fp" else lib().trh_best_fft_fq
    _check(fn(_p(a), _p(w), log_n))
    return a


def point_sum(curve: str, points) -> np.ndarray:
    pts = _c(points, 12)
    out = np.zeros(12, dtype=np.uint64)
    _check(lib().trh_point_sum(CURVE_ID[curve], _p(pts), pts.shape[0], _p(out)))
    return out


def affine_of(jac) -> np.ndarray:
    """Normalised Jacobian (Z = 1 or identity) -> 8-limb affine POD."""
    return np.ascontiguousarray(jac, dtype=np.uint64).reshape(12)[:8].copy()


# ---------------------------------------------------------------------------------------
# device memory helpers
# ---------------------------------------------------------------------------------------
class DeviceBuffer:
    def __init__(self, nbytes: int):
        self.ptr = _vp()
        self.nbytes = nbytes
        _check(lib().trh_malloc(ctypes.byref(self.ptr), nbytes))

    @classmethod
    def from_host(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        if a.nbytes:
            _check(lib().trh_memcpy_h2d(buf.ptr, a.ctypes.data_as(_vp), a.nbytes))
        return buf

    def to_host(self, dtype=np.uint64, shape=None) -> np.ndarray:
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if self.nbytes:
            _check(lib().trh_memcpy_d2h(out.ctypes.data_as(_vp), self.ptr, self.nbytes))
        return out if shape is None else out.reshape(shape)

    def free(self):
        if self.ptr:
            lib().trh_free(self.ptr)
            self.ptr = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _devptr(x):
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return _vp(x.data_ptr())
    return _vp(int(x))


def field_op_dev(field: str, op: str, a, b=None) -> np.ndarray:
    a = _c(a, 4)
    da = DeviceBuffer.from_host(a)
    db = DeviceBuffer.from_host(_c(b, 4)) if b is not None else None
    do = DeviceBuffer(a.nbytes)
    _check(lib().trh_field_op_dev(FIELD_ID[field], FIELD_OPS[op], da.ptr, db.ptr if db else None, do.ptr, a.shape[0], None))
    _check(lib().trh_stream_synchronize(None))
    return do.to_host(shape=(-1, 4))


def point_op_dev(curve: str, op: str, p, q=None) -> np.ndarray:
    p = _c(p, 12)
    dp = DeviceBuffer.from_host(p)
    dq = DeviceBuffer.from_host(_c(q)) if q is not None else None
    do = DeviceBuffer(p.nbytes)
    _check(lib().trh_point_op_dev(CURVE_ID[curve], POINT_OPS[op], dp.ptr, dq.ptr if dq else None, do.ptr, p.shape[0], None))
    _check(lib().trh_stream_synchronize(None))
    return do.to_host(shape=(-1, 12))


def ntt_dev(field: str, a_dev, log_n: int, omega, batch: int = 1, stream=None):
    w = _c(omega).reshape(4)
    _check(lib().trh_ntt_dev(FIELD_ID[field], _devptr(a_dev), log_n, _p(w), batch, stream))


def field_scale_dev(field: str, a_dev, n: int, factor, stream=None):
    f = _c(factor).reshape(4)
    _check(lib().trh_field_scale_dev(FIELD_ID[field], _devptr(a_dev), n, _p(f), stream))


def field_scale_periodic_dev(field: str, a_dev, n: int, factors, stream=None):
    f = _c(factors, 4)
    _check(lib().trh_field_scale_periodic_dev(FIELD_ID[field], _devptr(a_dev), n, _p(f), f.shape[0], stream))


def field_scale_rows_dev(field: str, a_dev, rows: int, row_len: int, active_len: int, factors, stream=None):
    f = _c(factors, 4)
    _check(lib().trh_field_scale_rows_dev(FIELD_ID[field], _devptr(a_dev), rows, row_len, active_len, _p(f), f.shape[0], stream))


def inner_product_dev(field: str, a_dev, b_dev, n: int, stream=None) -> np.ndarray:
    """halo2_proofs::arithmetic::compute_inner_product on device vectors -> (4,) Montgomery limbs"""
    out = np.zeros(4, dtype=np.uint64)
    _check(lib().trh_field_inner_product_dev(FIELD_ID[field], _devptr(a_dev), _devptr(b_dev), n, stream, _p(out)))
    return out


def poly_eval_batch_dev(field: str, polys_dev, n: int, batch: int, point, stream=None) -> np.ndarray:
    """arithmetic::eval_polynomial of `batch` device polynomials at one point -> (batch, 4) limbs"""
    out = np.zeros((batch, 4), dtype=np.uint64)
    x = _c(point).reshape(4)
    _check(lib().trh_poly_eval_batch_dev(FIELD_ID[field], _devptr(polys_dev), n, batch, _p(x), stream, _p(out)))
    return out


def axpy_dev(field: str, y_dev, x_dev, n: int, c, stream=None):
    cc = _c(c).reshape(4)
    _check(lib().trh_field_axpy_dev(FIELD_ID[field], _devptr(y_dev), _devptr(x_dev), n, _p(cc), stream))


def powers_dev(field: str, out_dev, n: int, x, stream=None):
    xx = _c(x).reshape(4)
    _check(lib().trh_field_powers_dev(FIELD_ID[field], _devptr(out_dev), n, _p(xx), stream))


def bases_fold_dev(curve: str, g_lo_dev, g_hi_dev, half: int, u, stream=None):
    uu = _c(u).reshape(4)
    _check(lib().trh_bases_fold_dev(CURVE_ID[curve], _devptr(g_lo_dev), _devptr(g_hi_dev), half, _p(uu), stream))


def point_fft_dev(curve: str, points_dev, log_n: int, omega, scale=None, stream=None):
    """best_fft over curve points (Params::new's g -> g_lagrange); `scale` multiplies every output (n^-1)"""
    w = _c(omega).reshape(4)
    sc = None if scale is None else _c(scale).reshape(4)
    _check(lib().trh_point_fft_dev(CURVE_ID[curve], _devptr(points_dev), log_n, _p(w), None if sc is None else _p(sc), stream))


def batch_invert_dev(field: str, a_dev, n: int, stream=None):
    """ff::BatchInvert on a device vector, in place (zeros stay zero)"""
    _check(lib().trh_field_batch_invert_dev(FIELD_ID[field], _devptr(a_dev), n, stream))


class ProductTerm(ctypes.Structure):
    """trh_product_term_t: x[i] + c * y[i] + g (y null: x[i] + g); c / g Montgomery words"""
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("c", ctypes.c_uint64 * 4), ("g", ctypes.c_uint64 * 4)]


def product_terms_dev(field: str, rows, n: int, out_dev, stream=None):
    """rows: a list of rows, each a list of terms (x_dev, y_dev or None, c limbs or None, g limbs); out[r][i] = prod of row r's terms
    at i -- the numerator / denominator products of every product column of a proof in one launch (trh_product_terms_dev)"""
    flat = [t for row in rows for t in row]
    arr = (ProductTerm * len(flat))()
    for d, (x, y, c, g) in zip(arr, flat):
        d.x = _devptr(x)
        d.y = _devptr(y) if y is not None else None
        if c is not None:
            d.c[:] = [int(v) for v in c]
        d.g[:] = [int(v) for v in g]
    starts = (ctypes.c_uint32 * (len(rows) + 1))()
    acc = 0
    for i, row in enumerate(rows):
        starts[i] = acc
        acc += len(row)
    starts[len(rows)] = acc
    _check(lib().trh_product_terms_dev(FIELD_ID[field], ctypes.cast(arr, ctypes.c_void_p), starts, len(rows), n, _devptr(out_dev), stream))


def batch_invert_mul_dev(field: str, a_dev, num_dev, n: int, stream=None):
    """a[i] <- num[i] / a[i] in place (a zero denominator stays zero): ff::BatchInvert and the multiply that follows it, one pass"""
    _check(lib().trh_field_batch_invert_mul_dev(FIELD_ID[field], _devptr(a_dev), _devptr(num_dev), n, stream))


def prefix_product_dev(field: str, a_dev, out_dev, n: int, stream=None):
    """out[i] = prod_{j < i} a[j] (out[0] = 1): the running product of the permutation / lookup z columns"""
    _check(lib().trh_field_prefix_product_dev(FIELD_ID[field], _devptr(a_dev), _devptr(out_dev), n, stream))


def set_timing(on: bool):
    _check(lib().trh_set_timing(1 if on else 0))


def last_timing() -> dict:
    t = Timing()
    _check(lib().trh_last_timing(ctypes.byref(t)))
    return {k: getattr(t, k) for k, _ in Timing._fields_}


def set_window_bits(c: int):
    _check(lib().trh_msm_set_window_bits(c))


# ---------------------------------------------------------------------------------------
# device-resident bases + MSM
# ---------------------------------------------------------------------------------------
class Bases:
    """Device-resident affine base set (Params.g / Params.g_lagrange)."""

    def __init__(self, curve: str, handle):
        self.curve = curve
        self.handle = handle

    @classmethod
    def from_host(cls, curve: str, xy) -> "Bases":
        xy = _c(xy, 8)
        h = _vp()
        fn = lib().trh_bases_create_pallas if curve == "pallas" else lib().trh_bases_create_vesta
        _check(fn(_p(xy), xy.shape[0], ctypes.byref(h)))
        return cls(curve, h)

    @classmethod
    def wrap_device(cls, curve: str, dev_ptr, n: int) -> "Bases":
        h = _vp()
        _check(lib().trh_bases_wrap_device(CURVE_ID[curve], _devptr(dev_ptr), n, ctypes.byref(h)))
        return cls(curve, h)

    @classmethod
    def generate(cls, curve: str, s0: int, d: int, n: int, first: int = 0) -> "Bases":
        h = _vp()
        _check(lib().trh_bases_generate(CURVE_ID[curve], s0, d, first, n, ctypes.byref(h)))
        return cls(curve, h)

    def __len__(self):
        return int(lib().trh_bases_len(self.handle))

    def shards(self) -> int:
        return int(lib().trh_bases_shards(self.handle))

    def precompute(self, window_bits: int = 0) -> int:
        """Attach the fixed-base table (2^(c j) P_i for all windows j); returns the window width used."""
        _check(lib().trh_bases_precompute(self.handle, window_bits))
        return int(lib().trh_bases_precomputed_window_bits(self.handle))

    def download(self, offset: int = 0, n: int | None = None) -> np.ndarray:
        n = len(self) - offset if n is None else n
        out = np.empty((n, 8), dtype=np.uint64)
        _check(lib().trh_bases_download(self.handle, offset, n, _p(out)))
        return out

    def msm(self, scalars, offset: int = 0, montgomery: bool = True) -> np.ndarray:
        s = _c(scalars, 4)
        out = np.zeros(12, dtype=np.uint64)
        _check(lib().trh_msm(self.handle, offset, _p(s), s.shape[0], 1 if montgomery else 0, _p(out)))
        return out

    def msm_dev(self, scalars_dev, n: int, offset: int = 0, montgomery: bool = True, stream=None) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        _check(lib().trh_msm_dev(self.handle, offset, _devptr(scalars_dev), n, 1 if montgomery else 0, stream, _p(out)))
        return out

    def msm_dev_enqueue(self, scalars_dev, n: int, offset: int = 0, montgomery: bool = True, stream=None):
        _check(lib().trh_msm_dev_enqueue(self.handle, offset, _devptr(scalars_dev), n, 1 if montgomery else 0, stream))

    def msm_dev_finish(self, stream=None) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        _check(lib().trh_msm_dev_finish(self.handle, stream, _p(out)))
        return out

    def msm_batch_dev(self, scalars_dev, n: int, batch: int, offset: int = 0, montgomery: bool = True, stream=None) -> np.ndarray:
        out = np.zeros((batch, 12), dtype=np.uint64)
        _check(lib().trh_msm_batch_dev(self.handle, offset, _devptr(scalars_dev), n, batch, 1 if montgomery else 0, stream, _p(out)))
        return out

    def commit_batch_dev(self, polys_dev, n: int, batch: int, blinds, stream=None) -> np.ndarray:
        """Params::commit(_lagrange) of `batch` device polynomials: MSM of polys[b] || blinds[b] over the n + 1 bases"""
        out = np.zeros((batch, 12), dtype=np.uint64)
        bl = _c(blinds, 4)
        assert bl.shape[0] == batch
        _check(lib().trh_commit_batch_dev(self.handle, _devptr(polys_dev), n, batch, _p(bl), stream, _p(out)))
        return out

    def commit_batch_host(self, polys, blinds) -> np.ndarray:
        """Params::commit(_lagrange) of a list of HOST polynomials ((n, 4) uint64 each): columns go up in chunks under the MSMs"""
        n = polys[0].shape[0]
        for a in polys:
            assert a.dtype == np.uint64 and a.flags.c_contiguous and a.shape == (n, 4)
        out = np.zeros((len(polys), 12), dtype=np.uint64)
        bl = _c(blinds, 4)
        assert bl.shape[0] == len(polys)
        _check(lib().trh_commit_batch_host(self.handle, _ptr_array(polys), n, len(polys), _p(bl), _p(out)))
        return out

    def destroy(self):
        if self.handle:
            lib().trh_bases_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
