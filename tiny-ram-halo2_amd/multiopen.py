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


def _limbs_many(field: str, values) -> np.ndarray:
    """(len, 4) Montgomery limbs of a list of integers (one bytes join instead of a numpy array per value: the x1 fold has 400 of them)"""
    m = _MODULUS[field]
    r = (1 << 256) % m
    buf = b"".join((v % m * r % m).to_bytes(32, "little") for v in values)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(values), 4).copy()


def _powers_desc(x: int, count: int, m: int):
    """[x^(count-1), ..., x, 1] by running products (pow() per entry costs a square-and-multiply chain each)"""
    out = [1] * count
    for j in range(count - 2, -1, -1):
        out[j] = out[j + 1] * x % m
    return out


def lincomb(field: str, polys, coeffs, out=None):
    """sum_b coeffs[b] * polys[b]; polys: device tensor (batch, n, 4), coeffs: ints"""
    import torch
    batch, n = polys.shape[0], polys.shape[1]
    assert len(coeffs) == batch and polys.is_contiguous()
    if out is None:
        out = torch.empty((n, 4), dtype=polys.dtype, device=polys.device)
    c = _limbs_many(field, coeffs)
    api._check(api.lib().trh_poly_lincomb_dev(api.FIELD_ID[field], api._devptr(polys), n, batch, api._p(c), api._devptr(out),
                                             torch.cuda.current_stream(polys.device).cuda_stream))
    return out


class KateDivider:
    """kate_division by (X - z) for polynomials of n coefficients; the powers of z and z^-1 are built once per point"""

    def __init__(self, field: str, n: int, z: int, device, share=None):
        """share: a KateDivider of the same point and at least n coefficients whose power tables are reused (their prefixes)"""
        import torch
        m = _MODULUS[field]
        self.field, self.n, self.z = field, n, z % m
        if self.z and share is not None:
            assert share.z == self.z and share.n >= n
            self.pz, self.pzinv, self.scratch = share.pz[:n], share.pzinv[:n], share.scratch[: 2 * n]
        elif self.z:
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


# ---------------------------------------------------------------------------------------
# poly::multiopen::create_proof -- the whole opening phase on resident polynomials
# ---------------------------------------------------------------------------------------
def construct_intermediate_sets(queries):
    """poly/multiopen.rs `construct_intermediate_sets` on (point, key) pairs (key identifies the polynomial):
    polynomials in order of first appearance, point indices in order of first appearance, point-index sets numbered in order
    of first appearance among the polynomials, the points of a set ordered by point index.
    Returns (commitments: list of (key, set_index), point_sets: list of lists of points)."""
    point_index, commitment_points, order = {}, {}, []
    for point, key in queries:
        idx = point_index.setdefault(point, len(point_index))
        if key not in commitment_points:
            commitment_points[key] = []
            order.append(key)
        commitment_points[key].append(idx)
    inverse = {i: p for p, i in point_index.items()}
    set_index, commitments = {}, []
    for key in order:
        s = tuple(sorted(set(commitment_points[key])))
        commitments.append((key, set_index.setdefault(s, len(set_index))))
    point_sets = [None] * len(set_index)
    for s, i in set_index.items():
        point_sets[i] = [inverse[j] for j in s]
    return commitments, point_sets


def _stack(tensors):
    """(batch, n, 4) view of the polynomials of one point set: in place when they are consecutive rows of one buffer (the prover keeps
    the coefficient forms of a set back to back), a copy otherwise"""
    import torch
    t0 = tensors[0]
    step = t0.numel() * t0.element_size()
    if all(t.is_contiguous() and t.data_ptr() == t0.data_ptr() + i * step and t.shape == t0.shape for i, t in enumerate(tensors)):
        try:
            return torch.as_strided(t0, (len(tensors),) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride()))
        except RuntimeError:
            pass  # the rows belong to different allocations that merely happen to be adjacent
    return torch.stack(tensors).contiguous()


def create_proof(params, rng, transcript, queries, polys: dict, blinds: dict, s_poly=None):
    """queries: list of (point, key) in the prover's order; polys[key]: device tensor (n, 4) of coefficients; blinds[key]: int.
    Mirrors poly/multiopen/prover.rs: x1 / x2 squeezes, per-set Horner fold in x1, kate_division by every point of the set,
    x2 fold, commitment of q'(X), x3, the evaluations of the q_i at x3, x4 fold and the IPA opening at x3.  Returns what the
    IPA returns; everything the Rust prover writes goes to `transcript` in the same order."""
    import torch
    from . import ipa
    curve, n = params.curve, params.n
    sf = api.SCALAR_FIELD[curve]
    m = _MODULUS[sf]
    x1 = transcript.squeeze_challenge_scalar()
    x2 = transcript.squeeze_challenge_scalar()
    commitments, point_sets = construct_intermediate_sets(queries)
    nsets = len(point_sets)
    members = [[key for key, s in commitments if s == i] for i in range(nsets)]
    q_polys, q_blinds = [], []
    for keys in members:  # q = (((p_0 x1 + p_1) x1 + p_2) ...): coefficient of p_j is x1^(len - 1 - j)
        coeffs = _powers_desc(x1, len(keys), m)
        stack = _stack([polys[k] for k in keys])
        q_polys.append(lincomb(sf, stack, coeffs))
        q_blinds.append(sum(c * blinds[k] for c, k in zip(coeffs, keys)) % m)
    dev = q_polys[0].device
    divided = []
    dividers = {}  # per point: the tables of z^i / z^-i at full length serve every later, shorter division by the same point
    for pts, q in zip(point_sets, q_polys):
        cur = q
        for z in pts:
            have = dividers.get(z % m)
            if have is not None and have.n < cur.shape[0]:
                have = None  # a longer polynomial than the tables reach: build them anew
            kd = KateDivider(sf, cur.shape[0], z, dev, share=have)
            if have is None:
                dividers[z % m] = kd
            cur = kd.divide(cur)
        pad = torch.zeros((n, 4), dtype=q.dtype, device=dev)   # poly.resize(params.n, 0)
        pad[: cur.shape[0]] = cur
        divided.append(pad)
    q_prime = lincomb(sf, torch.stack(divided).contiguous(), _powers_desc(x2, nsets, m))
    q_prime_blind = rng()
    transcript.write_point(params.commit(q_prime, _limbs(sf, q_prime_blind)))
    x3 = transcript.squeeze_challenge_scalar()
    evals = api.poly_eval_batch_dev(sf, torch.stack(q_polys).contiguous(), n, nsets, _limbs(sf, x3))
    for e in evals:
        transcript.write_scalar(e)
    x4 = transcript.squeeze_challenge_scalar()
    # p = ((q' x4 + q_0) x4 + q_1) ...: q' gets x4^nsets, q_i gets x4^(nsets - 1 - i)
    p_poly = lincomb(sf, torch.stack([q_prime] + q_polys).contiguous(), [pow(x4, nsets - i, m) for i in range(nsets + 1)])
    p_blind = (q_prime_blind * pow(x4, nsets, m) + sum(pow(x4, nsets - 1 - i, m) * b for i, b in enumerate(q_blinds))) % m
    if s_poly is None:  # the prover's random s(X): n draws (a caller that times the phase hands the host array in)
        s_poly = np.stack([_limbs(sf, rng()) for _ in range(n)])
    return ipa.create_proof_native(params, rng, transcript, p_poly, p_blind, x3, s_poly, rng())
