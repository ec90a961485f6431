"""Device versions of the polynomial arithmetic in halo2_proofs 0.2.0 `poly/multiopen/prover.rs` (reached from create_proof,
/root/reference/src/test_utils.rs:41-49): for every point set the queried polynomials are folded with powers of x1
(`lincomb`), the interpolated evaluations are subtracted and the result is divided by (X - point) for each point of the set
(`arithmetic::kate_division`), the per-set quotients are folded with x2, and after the commitment of that f(X) the final
polynomial is another x4 fold -- all on coefficient forms that are already resident on the device."""
from __future__ import annotations

import numpy as np

from . import api
from .poly import _MODULUS


def _limbs(field: str, v: int) -> np.ndarray:
    m = _MODULUS[field]
    x = v % m * ((1 << 256) % m) % m
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def lincomb(field: str, polys, coeffs, out=None):
    """sum_b coeffs[b] * polys[b]; polys: device tensor (batch, n, 4), coeffs: ints"""
    import torch
    batch, n = polys.shape[0], polys.shape[1]
    assert len(coeffs) == batch and polys.is_contiguous()
    if out is None:
        out = torch.empty((n, 4), dtype=polys.dtype, device=polys.device)
    c = np.stack([_limbs(field, v) for v in coeffs])
    api._check(api.lib().trh_poly_lincomb_dev(api.FIELD_ID[field], api._devptr(polys), n, batch, api._p(c), api._devptr(out),
                                             torch.cuda.current_stream(polys.device).cuda_stream))
    return out


class KateDivider:
    """kate_division by (X - z) for polynomials of n coefficients; the powers of z and z^-1 are built once per point"""

    def __init__(self, field: str, n: int, z: int, device):
        import torch
        m = _MODULUS[field]
        self.field, self.n, self.z = field, n, z % m
        if self.z:
            self.pz = torch.empty((n, 4), dtype=torch.int64, device=device)
            self.pzinv = torch.empty((n, 4), dtype=torch.int64, device=device)
            st = torch.cuda.current_stream(device).cuda_stream
            api.powers_dev(field, self.pz, n, _limbs(field, self.z), stream=st)
            api.powers_dev(field, self.pzinv, n, _limbs(field, pow(self.z, -1, m)), stream=st)
            self.scratch = torch.empty((2 * n, 4), dtype=torch.int64, device=device)

    def divide(self, a):
        """a: device tensor (n, 4) -> quotient (n - 1, 4); the remainder a(z) is dropped as in the Rust code"""
        import torch
        assert a.shape[0] == self.n and a.is_contiguous()
        if not self.z:  # division by X: shift
            return a[1:].clone()
        q = torch.empty((self.n - 1, 4), dtype=a.dtype, device=a.device)
        api._check(api.lib().trh_poly_kate_division_dev(api.FIELD_ID[self.field], api._devptr(a), self.n, api._devptr(self.pz), api._devptr(self.pzinv),
                                                       api._devptr(self.scratch), api._devptr(q), torch.cuda.current_stream(a.device).cuda_stream))
        return q
