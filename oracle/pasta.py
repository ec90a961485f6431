"""ORACLE (test infrastructure, NOT product code) -- Python big-int restatement of the
pasta_curves 0.4.1 field/curve arithmetic and of halo2_proofs 0.2.0
`arithmetic::{best_multiexp, best_fft}`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED BY THE REFERENCE: the hot path lives in two crates that are not vendored in
/root/reference (halo2_proofs 0.2.0 @ git Orbis-Tertius/halo2 rev a95945254dcc61acc1648c6039faeff85bc2440f,
Cargo.lock:619-621; pasta_curves 0.4.1, Cargo.lock:847-858) and there is no Rust toolchain
in this image, so the reference cannot be compiled or imported.  The reference's own tests
for this path (src/test_utils.rs:6-71, 73-119 -- prove/verify round trips with OsRng) hold
no value-level golden vectors.  What pins this oracle instead:
  * the published pasta_curves constants (modulus, R, R2, INV, ROOT_OF_UNITY, ZETA, DELTA,
    TWO_INV), re-derived here from first principles and asserted;
  * the unique mathematical definitions: field ops mod p/q, the group law of
    y^2 = x^3 + 5, MSM = sum_i s_i * P_i, DFT a'[i] = sum_j a[j] * omega^(i*j).
Field outputs are unique (fully reduced Montgomery limbs); group outputs are unique after
affine normalisation.  Three independent implementations are cross-checked in tests/:
this file (big-int), oracle/cpu_ref.cpp (4xu64 Montgomery), and the HIP path (8xu32).

Reference call sites these functions stand in for: src/test_utils.rs:21 (Params::new),
:23-25 (keygen_vk/keygen_pk), :41-49 (create_proof -> best_multiexp / best_fft / IPA).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------------------
# Fields.  pasta_curves::Fp / Fq (pasta_curves 0.4.1 src/fields/{fp,fq}.rs; used by the
# reference at src/test_utils.rs:2, src/circuits/tables/even_bits.rs:250-262).
# --------------------------------------------------------------------------------------
P_MOD = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q_MOD = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
MONT_R_BITS = 256
MASK64 = (1 << 64) - 1


@dataclass(frozen=True)
class Field:
    name: str
    m: int

    @property
    def R(self) -> int:  # Montgomery radix mod m
        return (1 << MONT_R_BITS) % self.m

    @property
    def R2(self) -> int:
        return pow(1 << MONT_R_BITS, 2, self.m)

    @property
    def R3(self) -> int:
        return pow(1 << MONT_R_BITS, 3, self.m)

    @property
    def INV(self) -> int:  # -m^{-1} mod 2^64
        return (-pow(self.m, -1, 1 << 64)) % (1 << 64)

    S = 32  # two-adicity of both fields
    GENERATOR = 5  # multiplicative generator (quadratic non-residue) of both fields

    @property
    def ROOT_OF_UNITY(self) -> int:  # primitive 2^32-th root: GENERATOR^((m-1)/2^32)
        return pow(self.GENERATOR, (self.m - 1) >> self.S, self.m)

    @property
    def DELTA(self) -> int:  # GENERATOR^(2^S)
        return pow(self.GENERATOR, 1 << self.S, self.m)

    @property
    def TWO_INV(self) -> int:
        return pow(2, -1, self.m)

    @property
    def ZETA(self) -> int:
        # cube root of unity used by halo2's extended coset; published values:
        # Fp: 5^(2(p-1)/3), Fq: 5^((q-1)/3)  
        e = (self.m - 1) // 3
        return pow(self.GENERATOR, 2 * e if self.name == "fp" else e, self.m)

    # ---- canonical-domain ops ----
    def add(self, a, b): return (a + b) % self.m
    def sub(self, a, b): return (a - b) % self.m
    def neg(self, a): return (-a) % self.m
    def mul(self, a, b): return (a * b) % self.m
    def sqr(self, a): return (a * a) % self.m
    def inv(self, a): return pow(a, -1, self.m) if a % self.m else 0  # pasta: invert(0) -> None; we map to 0

    def omega(self, log_n: int) -> int:
        """Domain generator halo2 uses: ROOT_OF_UNITY^(2^(S-log_n))."""
        assert 0 <= log_n <= self.S
        return pow(self.ROOT_OF_UNITY, 1 << (self.S - log_n), self.m)

    # ---- Montgomery encoding (memory format of pasta Fp/Fq: [u64;4] LE, value*R mod m) ----
    def to_mont(self, a: int) -> int: return (a * self.R) % self.m
    def from_mont(self, am: int) -> int: return (am * pow(self.R, -1, self.m)) % self.m

    def limbs(self, a: int) -> List[int]:
        """canonical int -> 4 u64 LE Montgomery limbs"""
        return int_to_limbs(self.to_mont(a % self.m))

    def from_limbs(self, l: Sequence[int]) -> int:
        return self.from_mont(limbs_to_int(l))

    def sqrt(self, a: int) -> Optional[int]:
        """Tonelli-Shanks (two-adicity 32)."""
        a %= self.m
        if a == 0:
            return 0
        if pow(a, (self.m - 1) // 2, self.m) != 1:
            return None
        q, s = self.m - 1, 0
        while q % 2 == 0:
            q //= 2; s += 1
        z = self.GENERATOR
        mm, c, t, r = s, pow(z, q, self.m), pow(a, q, self.m), pow(a, (q + 1) // 2, self.m)
        while t != 1:
            i, t2 = 0, t
            while t2 != 1:
                t2 = t2 * t2 % self.m; i += 1
            b = pow(c, 1 << (mm - i - 1), self.m)
            mm, c = i, b * b % self.m
            t, r = t * c % self.m, r * b % self.m
        return r


def int_to_limbs(x: int) -> List[int]:
    return [(x >> (64 * i)) & MASK64 for i in range(4)]


def limbs_to_int(l: Sequence[int]) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


FP = Field("fp", P_MOD)
FQ = Field("fq", Q_MOD)
FIELDS = {"fp": FP, "fq": FQ}

# published pasta_curves constants -- the oracle's pin.
PUBLISHED = {
    "fp": dict(
        INV=0x992D30ECFFFFFFFF,
        R=[0x34786D38FFFFFFFD, 0x992C350BE41914AD, 0xFFFFFFFFFFFFFFFF, 0x3FFFFFFFFFFFFFFF],
        R2=[0x8C78ECB30000000F, 0xD7D30DBD8B0DE0E7, 0x7797A99BC3C95D18, 0x096D41AF7B9CB714],
        R3=[0xF185A5993A9E10F9, 0xF6A68F3B6AC5B1D1, 0xDF8D1014353FD42C, 0x2AE309222D2D9910],
        ROOT_OF_UNITY=0x2BCE74DEAC30EBDA362120830561F81AEA322BF2B7BB7584BDAD6FABD87EA32F,
        ZETA=0x12CCCA834ACDBA712CAAD5DC57AAB1B01D1F8BD237AD31491DAD5EBDFDFE4AB9,
        TWO_INV=0x2000000000000000000000000000000011234C7E04A67C8DCC96987680000001,
        DELTA=0x0A757D0F0006AB6CBD455B7112A5049DF5E4F3F13EEE56366A6CCD20DD7B9BA2,
    ),
    "fq": dict(
        INV=0x8C46EB20FFFFFFFF,
        R=[0x5B2B3E9CFFFFFFFD, 0x992C350BE3420567, 0xFFFFFFFFFFFFFFFF, 0x3FFFFFFFFFFFFFFF],
        R2=[0xFC9678FF0000000F, 0x67BB433D891A16E3, 0x7FAE231004CCF590, 0x096D41AF7CCFDAA9],
        R3=[0x008B421C249DAE4C, 0xE13BDA50DBA41326, 0x88FECECB8E15CB63, 0x07DD97A06E6792C8],
        ROOT_OF_UNITY=0x2DE6A9B8746D3F589E5C4DFD492AE26E9BB97EA3C106F049A70E2C1102B6D05F,
        ZETA=0x06819A58283E528E511DB4D81CF70F5A0FED467D47C033AF2AA9D2E050AA0E4F,
        TWO_INV=0x2000000000000000000000000000000011234C7E04CA546EC623759080000001,
        DELTA=0x2237D5442372416606F0A88E7F7949F8E3AC3376541D11408494392472D1683C,
    ),
}


def check_published_constants() -> None:
    """Pin: every derived constant must equal the published pasta_curves value."""
    for name, f in FIELDS.items():
        pub = PUBLISHED[name]
        assert f.INV == pub["INV"], name
        assert int_to_limbs(f.R) == pub["R"], name
        assert int_to_limbs(f.R2) == pub["R2"], name
        assert int_to_limbs(f.R3) == pub["R3"], name
        assert f.ROOT_OF_UNITY == pub["ROOT_OF_UNITY"], name
        assert f.ZETA == pub["ZETA"], name
        assert f.TWO_INV == pub["TWO_INV"], name
        assert f.DELTA == pub["DELTA"], name
        assert pow(f.ROOT_OF_UNITY, 1 << 32, f.m) == 1 and pow(f.ROOT_OF_UNITY, 1 << 31, f.m) != 1
        assert pow(f.ZETA, 3, f.m) == 1 and f.ZETA != 1


# --------------------------------------------------------------------------------------
# Curves.  pasta_curves::{pallas, vesta} (src/curves.rs): y^2 = x^3 + 5, generator (-1, 2).
# Pallas: base Fp, scalar Fq (EpAffine).  Vesta: base Fq, scalar Fp (EqAffine -- the curve the
# reference proves over, src/test_utils.rs:12, 21).
# Affine points are (x, y) canonical ints; identity is None.
# --------------------------------------------------------------------------------------
Affine = Optional[Tuple[int, int]]


@dataclass(frozen=True)
class Curve:
    name: str
    base: Field
    scalar: Field
    b: int = 5

    @property
    def generator(self) -> Tuple[int, int]:
        return (self.base.m - 1, 2)

    def is_on_curve(self, pt: Affine) -> bool:
        if pt is None:
            return True
        x, y = pt
        return (y * y - x * x * x - self.b) % self.base.m == 0

    def neg(self, pt: Affine) -> Affine:
        return None if pt is None else (pt[0], (-pt[1]) % self.base.m)

    def add(self, p1: Affine, p2: Affine) -> Affine:
        m = self.base.m
        if p1 is None: return p2
        if p2 is None: return p1
        x1, y1 = p1; x2, y2 = p2
        if x1 == x2:
            if (y1 + y2) % m == 0:
                return None
            lam = (3 * x1 * x1) * pow(2 * y1, -1, m) % m
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, m) % m
        x3 = (lam * lam - x1 - x2) % m
        return (x3, (lam * (x1 - x3) - y1) % m)

    def double(self, p: Affine) -> Affine:
        return self.add(p, p)

    def mul(self, k: int, pt: Affine) -> Affine:
        k %= self.scalar.m
        acc: Affine = None
        while k:
            if k & 1:
                acc = self.add(acc, pt)
            pt = self.add(pt, pt)
            k >>= 1
        return acc

    def msm_naive(self, scalars: Sequence[int], bases: Sequence[Affine]) -> Affine:
        """Definition: sum_i scalars[i] * bases[i]."""
        assert len(scalars) == len(bases)
        acc: Affine = None
        for s, b in zip(scalars, bases):
            acc = self.add(acc, self.mul(s, b))
        return acc

    def lift_x(self, x: int) -> Affine:
        y = self.base.sqrt((x * x * x + self.b) % self.base.m)
        return None if y is None else (x, y)

    # 64-byte POD the C ABI uses for affine bases: x[4], y[4] Montgomery limbs; identity = all 0
    def affine_limbs(self, pt: Affine) -> List[int]:
        if pt is None:
            return [0] * 8
        return self.base.limbs(pt[0]) + self.base.limbs(pt[1])

    def affine_from_limbs(self, l: Sequence[int]) -> Affine:
        if all(int(v) == 0 for v in l[:8]):
            return None
        return (self.base.from_limbs(l[0:4]), self.base.from_limbs(l[4:8]))

    def jacobian_limbs_to_affine(self, l: Sequence[int]) -> Affine:
        """(X, Y, Z) Montgomery limbs -> affine (X/Z^2, Y/Z^3); Z = 0 is the identity."""
        f = self.base
        X, Y, Z = (f.from_limbs(l[0:4]), f.from_limbs(l[4:8]), f.from_limbs(l[8:12]))
        if Z == 0:
            return None
        zi = f.inv(Z)
        return (X * zi * zi % f.m, Y * zi * zi * zi % f.m)


PALLAS = Curve("pallas", FP, FQ)
VESTA = Curve("vesta", FQ, FP)
CURVES = {"pallas": PALLAS, "vesta": VESTA}


# --------------------------------------------------------------------------------------
# best_multiexp restatement (halo2_proofs 0.2.0 src/arithmetic.rs, multiexp_serial /
# best_multiexp; reached via src/test_utils.rs:41-49).
# Pure-Python loops: small cases only.
# --------------------------------------------------------------------------------------
def multiexp_window(n: int) -> int:
    """c = 1 if n<4, 3 if n<32, else ceil(ln n)."""
    if n < 4:
        return 1
    if n < 32:
        return 3
    return int(math.ceil(math.log(n)))


def multiexp_serial(curve: Curve, coeffs: Sequence[int], bases: Sequence[Affine], acc: Affine = None) -> Affine:
    assert len(coeffs) == len(bases)
    n = len(coeffs)
    c = multiexp_window(n)
    segments = 256 // c + 1
    reprs = [(s % curve.scalar.m).to_bytes(32, "little") for s in coeffs]  # to_repr()

    def get_at(segment: int, rep: bytes) -> int:
        skip_bits = segment * c
        skip_bytes = skip_bits // 8
        if skip_bytes >= 32:
            return 0
        v = int.from_bytes(rep[skip_bytes:skip_bytes + 8].ljust(8, b"\0"), "little")
        return (v >> (skip_bits - skip_bytes * 8)) % (1 << c)

    for seg in reversed(range(segments)):
        for _ in range(c):
            acc = curve.double(acc)
        buckets: List[Affine] = [None] * ((1 << c) - 1)
        for rep, base in zip(reprs, bases):
            d = get_at(seg, rep)
            if d:
                buckets[d - 1] = curve.add(buckets[d - 1], base)
        running: Affine = None
        for b in reversed(buckets):
            running = curve.add(running, b)
            acc = curve.add(acc, running)
    return acc


def best_multiexp(curve: Curve, coeffs: Sequence[int], bases: Sequence[Affine], threads: int = 1) -> Affine:
    """chunk-per-thread split, then sum (arithmetic.rs best_multiexp)."""
    assert len(coeffs) == len(bases)  # reference: assert_eq!(coeffs.len(), bases.len())
    n = len(coeffs)
    if n > threads and threads > 1:
        chunk = n // threads
        acc: Affine = None
        for lo in range(0, n, chunk):
            acc = curve.add(acc, multiexp_serial(curve, coeffs[lo:lo + chunk], bases[lo:lo + chunk]))
        return acc
    return multiexp_serial(curve, coeffs, bases)


# --------------------------------------------------------------------------------------
# best_fft restatement (arithmetic.rs best_fft: bit-reverse, twiddle scan, radix-2 DIT;
# natural order in and out) + the O(n^2) definition.
# --------------------------------------------------------------------------------------
def bitreverse(n: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def best_fft(field: Field, a: List[int], omega: int, log_n: int) -> List[int]:
    n = len(a)
    assert n == 1 << log_n  # reference: assert_eq!(a.len(), 1 << log_n)
    m = field.m
    a = list(a)
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw = [1] * max(n // 2, 1)
    for i in range(1, n // 2):
        tw[i] = tw[i - 1] * omega % m
    half, step = 1, n // 2
    while half < n:
        for start in range(0, n, 2 * half):
            for j in range(half):
                t = a[start + half + j] * tw[j * step] % m
                u = a[start + j]
                a[start + j] = (u + t) % m
                a[start + half + j] = (u - t) % m
        half *= 2
        step //= 2
    return a


def dft_naive(field: Field, a: Sequence[int], omega: int) -> List[int]:
    """Definition: a'[i] = sum_j a[j] * omega^(i*j)."""
    n, m = len(a), field.m
    return [sum(a[j] * pow(omega, i * j, m) for j in range(n)) % m for i in range(n)]


# --------------------------------------------------------------------------------------
# Deterministic synthetic inputs shared by tests / bench.
# SplitMix64 counter stream: word k of the stream = splitmix64(seed + (k+1)*GAMMA).
# --------------------------------------------------------------------------------------
GAMMA = 0x9E3779B97F4A7C15


def splitmix64_at(seed: int, k: int) -> int:
    z = (seed + (k + 1) * GAMMA) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def synth_scalar_limbs(seed: int, i: int) -> List[int]:
    """Scalar i of the synthetic stream: four u64 draws, top two bits cleared (< 2^254 < p, q).
    The 4 limbs are taken AS the in-memory representation (so they are a valid Montgomery
    residue of *some* field element and equally a valid canonical one)."""
    l = [splitmix64_at(seed, 4 * i + k) for k in range(4)]
    l[3] &= 0x3FFFFFFFFFFFFFFF
    return l


def synth_base(curve: Curve, s0: int, d: int, i: int) -> Affine:
    """Base i of the synthetic set: (s0 + i*d) * G  (known discrete logs)."""
    return curve.mul(s0 + i * d, curve.generator)


# --------------------------------------------------------------------------------------
# poly::EvaluationDomain restatement (halo2_proofs 0.2.0 src/poly/domain.rs; SURVEY.md section
# 8 row a5; reached via keygen_* / create_proof at src/test_utils.rs:23-25, 41-49).
# Values are canonical ints; vectors are Python lists.
# --------------------------------------------------------------------------------------
class EvaluationDomain:
    def __init__(self, field: Field, j: int, k: int):
        m = field.m
        self.field, self.k, self.n = field, k, 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        self.extended_omega = field.omega(ek)
        self.omega = field.omega(k)
        assert pow(self.extended_omega, 1 << (ek - k), m) == self.omega
        self.omega_inv = field.inv(self.omega)
        self.extended_omega_inv = field.inv(self.extended_omega)
        self.g_coset = field.ZETA
        self.g_coset_inv = field.sqr(field.ZETA)
        self.ifft_divisor = field.inv(1 << k)
        self.extended_ifft_divisor = field.inv(1 << ek)
        # t(X) = X^n - 1 evaluated on the coset zeta * extended_omega^i, inverted
        self.t_evaluations = [field.inv((pow(self.g_coset * pow(self.extended_omega, i, m) % m, self.n, m) - 1) % m)
                              for i in range(1 << (ek - k))]

    def extended_len(self) -> int:
        return 1 << self.extended_k

    def lagrange_to_coeff(self, a: List[int]) -> List[int]:
        assert len(a) == self.n
        m = self.field.m
        return [v * self.ifft_divisor % m for v in best_fft(self.field, a, self.omega_inv, self.k)]

    def _distribute_powers_zeta(self, a: List[int], into_coset: bool) -> List[int]:
        powers = [self.g_coset, self.g_coset_inv] if into_coset else [self.g_coset_inv, self.g_coset]
        m = self.field.m
        return [v if i % 3 == 0 else v * powers[i % 3 - 1] % m for i, v in enumerate(a)]

    def coeff_to_extended(self, a: List[int]) -> List[int]:
        assert len(a) == self.n
        a = self._distribute_powers_zeta(list(a), True) + [0] * (self.extended_len() - self.n)
        return best_fft(self.field, a, self.extended_omega, self.extended_k)

    def extended_to_coeff(self, a: List[int]) -> List[int]:
        assert len(a) == self.extended_len()
        m = self.field.m
        a = [v * self.extended_ifft_divisor % m for v in best_fft(self.field, a, self.extended_omega_inv, self.extended_k)]
        a = self._distribute_powers_zeta(a, False)
        return a[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a: List[int]) -> List[int]:
        assert len(a) == self.extended_len()
        t, m = self.t_evaluations, self.field.m
        return [v * t[i % len(t)] % m for i, v in enumerate(a)]


# --------------------------------------------------------------------------------------
# best_fft over curve points (arithmetic.rs best_fft::<C::Curve>, as Params::new builds g_lagrange:
# g_lagrange = batch_normalize(n^-1 * best_fft(g, omega^-1, k)); src/test_utils.rs:21, 89).
# --------------------------------------------------------------------------------------
def best_fft_points(curve: Curve, a: List[Affine], omega: int, log_n: int) -> List[Affine]:
    n = len(a)
    assert n == 1 << log_n
    m = curve.scalar.m
    a = list(a)
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    half = 1
    while half < n:
        w_m = pow(omega, n // (2 * half), m)
        for start in range(0, n, 2 * half):
            w = 1
            for j in range(half):
                t = curve.mul(w, a[start + half + j])
                u = a[start + j]
                a[start + j] = curve.add(u, t)
                a[start + half + j] = curve.add(u, curve.neg(t))
                w = w * w_m % m
        half *= 2
    return a


def params_g_lagrange(curve: Curve, g: List[Affine], k: int) -> List[Affine]:
    """the Lagrange-basis generators Params::new derives from g"""
    f = curve.scalar
    n_inv = f.inv(1 << k)
    return [curve.mul(n_inv, p) for p in best_fft_points(curve, g, f.inv(f.omega(k)), k)]


# --------------------------------------------------------------------------------------
# IPA opening restatement (halo2_proofs 0.2.0 src/poly/commitment/prover.rs create_proof;
# SURVEY.md section 8 row a7; reached via create_proof at src/test_utils.rs:41-49).
# Transcript and randomness are injected (the reference feeds OsRng and a BLAKE2b transcript).
# --------------------------------------------------------------------------------------
def ipa_create_proof(curve: Curve, k: int, g: List[Affine], w: Affine, u: Affine, rng, transcript,
                     p_poly: List[int], p_blind: int, x3: int, s_poly: List[int], s_blind: int):
    f_, m, n = curve.scalar, curve.scalar.m, 1 << k
    assert len(p_poly) == n and len(g) == n and len(s_poly) == n

    def commit(poly, r):
        return best_multiexp(curve, list(poly) + [r], list(g) + [w])

    def eval_poly(poly, x):
        acc = 0
        for c in reversed(poly):
            acc = (acc * x + c) % m
        return acc

    s_poly = list(s_poly)
    s_poly[0] = (s_poly[0] - eval_poly(s_poly, x3)) % m      # s(x3) = 0
    transcript.write_point(commit(s_poly, s_blind))
    xi = transcript.squeeze_challenge_scalar()
    z = transcript.squeeze_challenge_scalar()
    p_prime = [(s * xi + p) % m for s, p in zip(s_poly, p_poly)]
    v = eval_poly(p_prime, x3)
    p_prime[0] = (p_prime[0] - v) % m
    f = (s_blind * xi + p_blind) % m
    b, cur = [], 1
    for _ in range(n):
        b.append(cur)
        cur = cur * x3 % m
    g_prime = list(g)
    for j in range(k):
        half = 1 << (k - j - 1)
        l_j = best_multiexp(curve, p_prime[half:], g_prime[:half])
        r_j = best_multiexp(curve, p_prime[:half], g_prime[half:])
        value_l = sum(x * y for x, y in zip(p_prime[half:], b[:half])) % m
        value_r = sum(x * y for x, y in zip(p_prime[:half], b[half:])) % m
        l_rand, r_rand = rng(), rng()
        l_j = curve.add(l_j, best_multiexp(curve, [value_l * z % m, l_rand], [u, w]))
        r_j = curve.add(r_j, best_multiexp(curve, [value_r * z % m, r_rand], [u, w]))
        transcript.write_point(l_j)
        transcript.write_point(r_j)
        u_j = transcript.squeeze_challenge_scalar()
        u_inv = f_.inv(u_j)
        for i in range(half):
            p_prime[i] = (p_prime[i] + p_prime[i + half] * u_inv) % m
            b[i] = (b[i] + b[i + half] * u_j) % m
        p_prime, b = p_prime[:half], b[:half]
        g_prime = [curve.add(g_prime[i], curve.mul(u_j, g_prime[i + half])) for i in range(half)]  # parallel_generator_collapse
        f = (f + l_rand * u_inv + r_rand * u_j) % m
    c = p_prime[0]
    transcript.write_scalar(c)
    transcript.write_scalar(f)
    return c, f


def ipa_verify_proof(curve: Curve, k: int, g: List[Affine], w: Affine, u: Affine, commitment: Affine, x3: int, v: int,
                     s_commitment: Affine, xi: int, z: int, rounds, challenges: List[int], c: int, f: int) -> bool:
    """halo2_proofs 0.2.0 poly/commitment/verifier.rs `verify_proof`, the equation the reference's own prove -> verify tests
    exercise (/root/reference/src/test_utils.rs:52-68, 106-118):

        P - [v] G_0 + [xi] S + sum_j [u_j^-1] L_j + sum_j [u_j] R_j  ==  [c] G'_0 + [c b z] U + [f] W

    with G'_0 = sum_i s_i G_i and b = sum_i s_i x3^i, s_i = prod over the rounds j whose fold put index i in the upper half
    of u_j (`compute_s` / `compute_b`).  Independent of the prover restatement above: it only uses the group law."""
    f_, m, n = curve.scalar, curve.scalar.m, 1 << k
    assert len(g) == n and len(rounds) == k and len(challenges) == k
    s = [1] * n
    for j, u_j in enumerate(challenges):
        bit = k - 1 - j  # round j folds index i + half (half = 2^(k-j-1)) onto i with weight u_j
        for i in range(n):
            if (i >> bit) & 1:
                s[i] = s[i] * u_j % m
    b = 0
    for i in range(n):
        b = (b + s[i] * pow(x3, i, m)) % m
    lhs = curve.add(commitment, curve.mul((-v) % m, g[0]))
    lhs = curve.add(lhs, curve.mul(xi, s_commitment))
    for (l_j, r_j), u_j in zip(rounds, challenges):
        lhs = curve.add(lhs, curve.mul(f_.inv(u_j), l_j))
        lhs = curve.add(lhs, curve.mul(u_j, r_j))
    g0 = best_multiexp(curve, s, list(g))
    rhs = curve.add(curve.mul(c, g0), curve.mul(c * b % m * z % m, u))
    rhs = curve.add(rhs, curve.mul(f, w))
    return lhs == rhs


if __name__ == "__main__":
    check_published_constants()
    for c in CURVES.values():
        assert c.is_on_curve(c.generator)
        assert c.mul(c.scalar.m, c.generator) is None
    print("oracle constants OK")


# ---------------------------------------------------------------------------------------
# Gate-expression evaluation (halo2_proofs 0.2.0 plonk/circuit.rs `Expression::evaluate`, as create_proof uses it
# for the quotient numerator: /root/reference/src/test_utils.rs:41-49; gates of the reference:
# src/circuits/tables/exe.rs:147-498).  Plain restatement over ints; nodes are tuples:
#   ("const", v) | ("col", key, rotation) | ("neg", e) | ("sum", a, b) | ("prod", a, b) | ("scaled", e, v)
# ---------------------------------------------------------------------------------------
def evaluate_expression(f, node, columns, row, n, rot_step):
    tag = node[0]
    if tag == "const":
        return node[1] % f.m
    if tag == "col":
        return columns[node[1]][(row + node[2] * rot_step) % n]
    if tag == "neg":
        return (-evaluate_expression(f, node[1], columns, row, n, rot_step)) % f.m
    if tag == "sum":
        return (evaluate_expression(f, node[1], columns, row, n, rot_step) + evaluate_expression(f, node[2], columns, row, n, rot_step)) % f.m
    if tag == "prod":
        return evaluate_expression(f, node[1], columns, row, n, rot_step) * evaluate_expression(f, node[2], columns, row, n, rot_step) % f.m
    if tag == "scaled":
        return evaluate_expression(f, node[1], columns, row, n, rot_step) * node[2] % f.m
    raise ValueError(tag)


def evaluate_gates(f, gates, columns, y, n, rot_step=1):
    """h numerator as create_proof folds it: acc = acc * y + gate, gate by gate, for every row"""
    out = []
    for row in range(n):
        acc = 0
        for g in gates:
            acc = (acc * y + evaluate_expression(f, g, columns, row, n, rot_step)) % f.m
        out.append(acc)
    return out


# ---------------------------------------------------------------------------------------
# Lookup argument: halo2_proofs 0.2.0 plonk/lookup/prover.rs `permute_expression_pair` (reached from create_proof,
# /root/reference/src/test_utils.rs:41-49), restated step by step: sort the input, walk it, first occurrences take
# their own value and consume one instance from the table's BTreeMap<value, count>, repeated rows are remembered;
# then the map is walked in ascending order and every left-over instance goes to `repeated_input_rows.pop()`.
# ---------------------------------------------------------------------------------------
def permute_expression_pair(input_values, table_values, usable_rows):
    permuted_input = sorted(input_values[:usable_rows])
    leftover = {}
    for v in table_values[:usable_rows]:
        leftover[v] = leftover.get(v, 0) + 1
    permuted_table = [0] * usable_rows
    repeated_input_rows = []
    for row, v in enumerate(permuted_input):
        if row == 0 or v != permuted_input[row - 1]:
            permuted_table[row] = v
            if leftover.get(v, 0) == 0:
                raise ValueError("ConstraintSystemFailure: input value not in the table")
            leftover[v] -= 1
        else:
            repeated_input_rows.append(row)
    for v in sorted(leftover):
        for _ in range(leftover[v]):
            permuted_table[repeated_input_rows.pop()] = v
    assert not repeated_input_rows
    return permuted_input, permuted_table


# ---------------------------------------------------------------------------------------
# poly::multiopen::create_proof (halo2_proofs 0.2.0 poly/multiopen/prover.rs + multiopen.rs construct_intermediate_sets,
# reached from create_proof: /root/reference/src/test_utils.rs:41-49), restated over integer coefficient lists.
# ---------------------------------------------------------------------------------------
def kate_division(f, a, z):
    q, tmp = [0] * (len(a) - 1), 0
    for i in range(len(a) - 1, 0, -1):
        tmp = (a[i] + z * tmp) % f.m
        q[i - 1] = tmp
    return q


def multiopen_create_proof(curve: Curve, k: int, g, w, u, rng, transcript, queries, polys, blinds):
    """queries: [(point, key)], polys[key]: n coefficients, blinds[key]: int.  Written to `transcript`: the commitment of q',
    the evaluations of the q_i at x3, then everything the IPA writes."""
    f_, m, n = curve.scalar, curve.scalar.m, 1 << k
    x1 = transcript.squeeze_challenge_scalar()
    x2 = transcript.squeeze_challenge_scalar()
    point_index, commitment_points, order = {}, {}, []
    for point, key in queries:
        idx = point_index.setdefault(point, len(point_index))
        if key not in commitment_points:
            commitment_points[key] = []
            order.append(key)
        commitment_points[key].append(idx)
    inverse = {i: p for p, i in point_index.items()}
    set_index, set_of = {}, {}
    for key in order:
        s_ = tuple(sorted(set(commitment_points[key])))
        set_of[key] = set_index.setdefault(s_, len(set_index))
    point_sets = [None] * len(set_index)
    for s_, i in set_index.items():
        point_sets[i] = [inverse[j] for j in s_]
    q_polys, q_blinds = [None] * len(point_sets), [0] * len(point_sets)
    for key in order:  # accumulate(set_idx, poly, blind): q = q * x1 + poly
        i = set_of[key]
        q_polys[i] = list(polys[key]) if q_polys[i] is None else [(a * x1 + b) % m for a, b in zip(q_polys[i], polys[key])]
        q_blinds[i] = (q_blinds[i] * x1 + blinds[key]) % m
    q_prime = None
    for pts, q in zip(point_sets, q_polys):
        cur = q
        for z in pts:
            cur = kate_division(f_, cur, z)
        cur = cur + [0] * (n - len(cur))
        q_prime = cur if q_prime is None else [(a * x2 + b) % m for a, b in zip(q_prime, cur)]
    q_prime_blind = rng()
    transcript.write_point(best_multiexp(curve, q_prime + [q_prime_blind], list(g) + [w]))
    x3 = transcript.squeeze_challenge_scalar()
    for q in q_polys:
        acc = 0
        for cf in reversed(q):
            acc = (acc * x3 + cf) % m
        transcript.write_scalar(acc)
    x4 = transcript.squeeze_challenge_scalar()
    p_poly, p_blind = q_prime, q_prime_blind
    for q, b in zip(q_polys, q_blinds):
        p_poly = [(a * x4 + c) % m for a, c in zip(p_poly, q)]
        p_blind = (p_blind * x4 + b) % m
    s_poly = [rng() for _ in range(n)]
    return ipa_create_proof(curve, k, g, w, u, rng, transcript, p_poly, p_blind, x3, s_poly, rng())
