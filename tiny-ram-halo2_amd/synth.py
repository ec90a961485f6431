"""Deterministic synthetic inputs for the MSM / NTT hot path.

Word k of stream `seed` is splitmix64(seed + (k+1)*GAMMA); a scalar / field element is four
consecutive words with the top two bits of the last limb cleared, so the 256-bit pattern is
< 2^254 < p, q and is a valid in-memory value of pasta Fp / Fq (it is used directly as the
Montgomery representation, which is what halo2's `&[C::Scalar]` slices hold).
"""
from __future__ import annotations

import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
SEED_MSM = 0x7472682D6D736D00  # "trh-msm\0" | log2(n)
SEED_NTT = 0x7472682D6E747400  # "trh-ntt\0" | log2(n)
# synthetic base set: P_i = (S0 + i*D) * G, G = (-1, 2)
BASE_S0 = 0x1234567
BASE_D = 0x89ABCDEF


def splitmix64_stream(seed: int, start: int, count: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def field_elements(seed: int, n: int, start: int = 0) -> np.ndarray:
    """(n, 4) uint64 limbs, each row < 2^254."""
    a = splitmix64_stream(seed, 4 * start, 4 * n).reshape(n, 4)
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


def msm_scalars(log_n: int, n: int | None = None, start: int = 0) -> np.ndarray:
    return field_elements(SEED_MSM | log_n, (1 << log_n) if n is None else n, start)


def ntt_input(log_n: int) -> np.ndarray:
    return field_elements(SEED_NTT | log_n, 1 << log_n)


def limbs_to_ints(a: np.ndarray) -> list:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


def ints_to_limbs(vals) -> np.ndarray:
    out = np.empty((len(vals), 4), dtype=np.uint64)
    m = (1 << 64) - 1
    for i, v in enumerate(vals):
        out[i] = [(v >> 0) & m, (v >> 64) & m, (v >> 128) & m, (v >> 192) & m]
    return out


def weighted_scalar_sum(scalars_canonical: np.ndarray, s0: int, d: int, start: int = 0) -> int:
    """sum_i scalar_i * (s0 + (start+i)*d) as an exact Python int (closed-form MSM check:
    expected MSM result = (that sum mod r) * G when bases are P_i = (s0 + i*d) G)."""
    s = np.asarray(scalars_canonical, dtype=np.uint64).reshape(-1, 4)
    n = s.shape[0]
    lo32 = s & np.uint64(0xFFFFFFFF)
    hi32 = s >> np.uint64(32)
    halves = np.stack([lo32[:, 0], hi32[:, 0], lo32[:, 1], hi32[:, 1],
                       lo32[:, 2], hi32[:, 2], lo32[:, 3], hi32[:, 3]], axis=1)
    total_s = 0      # sum_i scalar_i
    total_is = 0     # sum_i i * scalar_i
    blk = 1 << 16    # idx < 2^16, half-limb < 2^32: products < 2^48, block sums < 2^64
    for b0 in range(0, n, blk):
        h = halves[b0:b0 + blk]
        m = h.shape[0]
        col = h.sum(axis=0, dtype=np.uint64)
        idx = np.arange(m, dtype=np.uint64)
        wcol = (h * idx[:, None]).sum(axis=0, dtype=np.uint64)
        bs = sum(int(col[k]) << (32 * k) for k in range(8))
        bw = sum(int(wcol[k]) << (32 * k) for k in range(8))
        total_s += bs
        total_is += bw + (b0 + start) * bs
    return s0 * total_s + d * total_is
