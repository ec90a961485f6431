// Pasta field arithmetic for gfx950: 255-bit Montgomery (R = 2^256), nine 30-bit limbs in
// registers.
//
// Replaces pasta_curves 0.4.1 `Fp` / `Fq` (fields/fp.rs, fields/fq.rs; pinned at
// /root/reference/Cargo.lock:847-858, used by the reference at src/test_utils.rs:2 and
// src/circuits/tables/even_bits.rs:250-262).  The MEMORY format is identical to the Rust one
// (four u64 little-endian limbs, Montgomery form with R = 2^256, fully reduced), so buffers
// cross the C ABI without repacking; fe_load / fe_store convert between those eight 32-bit
// words and the register form.
//
// Why 30-bit limbs.  gfx950 has no 64x64 multiplier; its widest integer multiply is
// v_mad_u64_u32 (32x32 + 64 -> 64, half rate -- measured 32 Tops/s vs 65 for v_add_u32, the same
// rate as a carry-chained v_addc_co_u32).  With saturated 32-bit limbs every partial product
// needs carry handling and 64-bit operand pairs must be re-formed (the compiler emitted 211
// v_mov + 94 v_lshl_add_u64 around the 88 multiplies).  With limbs < 2^30 a column of the
// schoolbook product holds at most 9 products + 4 reduction terms < 13 * 2^60 < 2^64, so the
// whole multiply is `acc[i+j] += a[i] * b[j]` -- in-place v_mad_u64_u32 chains with no carries.
//
// Both moduli are m = 2^254 + t with t < 2^126 and m = 1 (mod 2^30): in radix 2^30 the limbs are
// [1, P1, P2, P3, P4, 0, 0, 0, 2^14].  The Montgomery quotient digit is q = -T_i mod 2^30 (no
// multiply), q*m costs four multiplies and one shift.  R stays 2^256 = 2^(8*30 + 16): eight
// 30-bit rounds, one 16-bit round, and a 16-bit realignment of the result.
//
// The same source is compiled for the host (final window combine, affine normalisation).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TRH_HD __host__ __device__ __forceinline__
#else
#define TRH_HD inline
#endif

namespace trh {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

constexpr u32 LIMB_BITS = 30;
constexpr u32 LIMB_MASK = (1u << LIMB_BITS) - 1u;
constexpr int NLIMBS = 9;

// 30-bit limb k of a 256-bit value given as eight 32-bit words (compile-time helper)
constexpr u32 limb30_of(const u32 (&w)[8], int k) {
    const int bit = 30 * k, word = bit >> 5, sh = bit & 31;
    u64 v = w[word];
    if (word + 1 < 8) v |= (u64)w[word + 1] << 32;
    return (u32)(v >> sh) & LIMB_MASK;
}

struct FpParams {  // Pallas base field = Vesta scalar field
    static constexpr int ID = 0;
    static constexpr u32 MOD[8] = {0x00000001u, 0x992d30edu, 0x094cf91bu, 0x224698fcu, 0u, 0u, 0u, 0x40000000u};
    static constexpr u32 ONE[8] = {0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};  // R mod p
    static constexpr u32 R2[8] = {0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu, 0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu};   // R^2 mod p
    // pasta_curves ROOT_OF_UNITY (primitive 2^32-th root) and ZETA (cube root of unity), Montgomery form
    static constexpr u32 ROOT_OF_UNITY[8] = {0xbad6dbf0u, 0xa28db849u, 0xd3b539dfu, 0x9083cd03u, 0x9dc8448eu, 0xfba6b9cau, 0x7b89c6dau, 0x3ec92874u};
    static constexpr u32 ZETA[8] = {0x619a153du, 0x02021cf6u, 0x4980b78eu, 0x9e8c2697u, 0xc87a4666u, 0x2a676d5cu, 0xa7a17876u, 0x15d8049du};
    static constexpr u32 TO_LAZY29[8] = {0xfffff001u, 0xc61e60ecu, 0x39bb3f88u, 0xb8b6d867u, 0xfffffddbu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};   // 2^266 mod p
    static constexpr u32 LAZY29_ONE[8] = {0xffffff81u, 0x0294ba6cu, 0x62d06b4fu, 0xfefa1af7u, 0xffffffeeu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};  // 2^261 mod p
};
struct FqParams {  // Vesta base field = Pallas scalar field
    static constexpr int ID = 1;
    static constexpr u32 MOD[8] = {0x00000001u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu, 0u, 0u, 0u, 0x40000000u};
    static constexpr u32 ONE[8] = {0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
    static constexpr u32 R2[8] = {0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu};
    static constexpr u32 ROOT_OF_UNITY[8] = {0x8c9942deu, 0x21807742u, 0x21b60494u, 0xcc495789u, 0xb2efbee2u, 0xac2e5d27u, 0x7f2db056u, 0x0b79fa89u};
    static constexpr u32 ZETA[8] = {0x80111122u, 0x7c541a84u, 0x56ed29dau, 0x40630b9cu, 0x135b2b29u, 0x02c275fbu, 0x88245b10u, 0x121d29f8u};
    static constexpr u32 TO_LAZY29[8] = {0xfffff001u, 0x1d94db20u, 0xbf06d019u, 0xb8b6d862u, 0xfffffddbu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};   // 2^266 mod q
    static constexpr u32 LAZY29_ONE[8] = {0xffffff81u, 0x68d15aa0u, 0x3f403a17u, 0xfefa1af7u, 0xffffffeeu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};  // 2^261 mod q
};

template <class F, int K> struct ModLimb { static constexpr u32 v = limb30_of(F::MOD, K); };
template <class F, int K> struct OneLimb { static constexpr u32 v = limb30_of(F::ONE, K); };
template <class F, int K> struct R2Limb { static constexpr u32 v = limb30_of(F::R2, K); };

template <class F> TRH_HD constexpr u32 mod_limb(int k) {
    return k == 0 ? ModLimb<F, 0>::v : k == 1 ? ModLimb<F, 1>::v : k == 2 ? ModLimb<F, 2>::v : k == 3 ? ModLimb<F, 3>::v :
           k == 4 ? ModLimb<F, 4>::v : k == 5 ? ModLimb<F, 5>::v : k == 6 ? ModLimb<F, 6>::v : k == 7 ? ModLimb<F, 7>::v : ModLimb<F, 8>::v;
}
template <class F> TRH_HD constexpr u32 one_limb(int k) {
    return k == 0 ? OneLimb<F, 0>::v : k == 1 ? OneLimb<F, 1>::v : k == 2 ? OneLimb<F, 2>::v : k == 3 ? OneLimb<F, 3>::v :
           k == 4 ? OneLimb<F, 4>::v : k == 5 ? OneLimb<F, 5>::v : k == 6 ? OneLimb<F, 6>::v : k == 7 ? OneLimb<F, 7>::v : OneLimb<F, 8>::v;
}
template <class F> TRH_HD constexpr u32 r2_limb(int k) {
    return k == 0 ? R2Limb<F, 0>::v : k == 1 ? R2Limb<F, 1>::v : k == 2 ? R2Limb<F, 2>::v : k == 3 ? R2Limb<F, 3>::v :
           k == 4 ? R2Limb<F, 4>::v : k == 5 ? R2Limb<F, 5>::v : k == 6 ? R2Limb<F, 6>::v : k == 7 ? R2Limb<F, 7>::v : R2Limb<F, 8>::v;
}
static_assert(ModLimb<FpParams, 0>::v == 1 && ModLimb<FpParams, 5>::v == 0 && ModLimb<FpParams, 6>::v == 0 &&
              ModLimb<FpParams, 7>::v == 0 && ModLimb<FpParams, 8>::v == (1u << 14), "Fp modulus shape");
static_assert(ModLimb<FqParams, 0>::v == 1 && ModLimb<FqParams, 5>::v == 0 && ModLimb<FqParams, 6>::v == 0 &&
              ModLimb<FqParams, 7>::v == 0 && ModLimb<FqParams, 8>::v == (1u << 14), "Fq modulus shape");

// register form: l[k] < 2^30, value = sum l[k] 2^(30k) < m
template <class F>
struct Fe {
    u32 l[NLIMBS];
};

// ---- memory <-> register form (eight 32-bit words, little endian) -----------------------
template <class F> TRH_HD Fe<F> fe_load(u32 w0, u32 w1, u32 w2, u32 w3, u32 w4, u32 w5, u32 w6, u32 w7) {
    Fe<F> r;
    r.l[0] = w0 & LIMB_MASK;
    r.l[1] = ((w0 >> 30) | (w1 << 2)) & LIMB_MASK;
    r.l[2] = ((w1 >> 28) | (w2 << 4)) & LIMB_MASK;
    r.l[3] = ((w2 >> 26) | (w3 << 6)) & LIMB_MASK;
    r.l[4] = ((w3 >> 24) | (w4 << 8)) & LIMB_MASK;
    r.l[5] = ((w4 >> 22) | (w5 << 10)) & LIMB_MASK;
    r.l[6] = ((w5 >> 20) | (w6 << 12)) & LIMB_MASK;
    r.l[7] = ((w6 >> 18) | (w7 << 14)) & LIMB_MASK;
    r.l[8] = w7 >> 16;
    return r;
}
template <class F> TRH_HD Fe<F> fe_load(const u32* w) { return fe_load<F>(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]); }
template <class F> TRH_HD void fe_store(const Fe<F>& a, u32* w) {
    w[0] = a.l[0] | (a.l[1] << 30);
    w[1] = (a.l[1] >> 2) | (a.l[2] << 28);
    w[2] = (a.l[2] >> 4) | (a.l[3] << 26);
    w[3] = (a.l[3] >> 6) | (a.l[4] << 24);
    w[4] = (a.l[4] >> 8) | (a.l[5] << 22);
    w[5] = (a.l[5] >> 10) | (a.l[6] << 20);
    w[6] = (a.l[6] >> 12) | (a.l[7] << 18);
    w[7] = (a.l[7] >> 14) | (a.l[8] << 16);
}

template <class F> TRH_HD Fe<F> fe_zero() {
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = 0;
    return r;
}
template <class F> TRH_HD Fe<F> fe_one() {
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = one_limb<F>(i);
    return r;
}
template <class F> TRH_HD Fe<F> fe_r2() {
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = r2_limb<F>(i);
    return r;
}
template <class F> TRH_HD bool fe_is_zero(const Fe<F>& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) o |= a.l[i];
    return o == 0;
}
template <class F> TRH_HD bool fe_eq(const Fe<F>& a, const Fe<F>& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// 2^14 in a register the compiler cannot see through: `q * two14 + acc` then stays ONE v_mad_u64_u32
// instead of a 64-bit shift plus a 64-bit add (the top modulus limb is 2^14)
TRH_HD u32 opaque_two14() {
    u32 v = 1u << 14;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// a in [0, 2m) with normalised limbs -> a mod m
template <class F> TRH_HD void fe_cond_sub(Fe<F>& a) {
    u32 t[NLIMBS];
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) {
        i32 v = (i32)a.l[i] - (i32)mod_limb<F>(i) + c;
        t[i] = (u32)v & LIMB_MASK;
        c = v >> 30;  // arithmetic: 0 or -1
    }
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) a.l[i] = c ? a.l[i] : t[i];
}

template <class F> TRH_HD Fe<F> fe_add(const Fe<F>& a, const Fe<F>& b) {
    Fe<F> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) {
        u32 v = a.l[i] + b.l[i] + c;
        r.l[i] = v & LIMB_MASK;
        c = v >> 30;
    }
    fe_cond_sub(r);  // a + b < 2m < 2^256 fits nine limbs
    return r;
}
template <class F> TRH_HD Fe<F> fe_sub(const Fe<F>& a, const Fe<F>& b) {
    Fe<F> r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) {
        i32 v = (i32)a.l[i] - (i32)b.l[i] + c;
        r.l[i] = (u32)v & LIMB_MASK;
        c = v >> 30;
    }
    const u32 mask = (u32)c;  // all ones when a < b: add m back
    u32 cc = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) {
        u32 v = r.l[i] + (mod_limb<F>(i) & mask) + cc;
        r.l[i] = v & LIMB_MASK;
        cc = v >> 30;
    }
    return r;
}
template <class F> TRH_HD Fe<F> fe_neg(const Fe<F>& a) { return fe_sub(fe_zero<F>(), a); }
template <class F> TRH_HD Fe<F> fe_dbl(const Fe<F>& a) { return fe_add(a, a); }

// Montgomery reduction of 18 lazy 64-bit columns (value < m * 2^256): returns value / 2^256 mod m.
template <class F> TRH_HD Fe<F> fe_mont_reduce(u64 (&acc)[18]) {
    constexpr u32 P1 = ModLimb<F, 1>::v, P2 = ModLimb<F, 2>::v, P3 = ModLimb<F, 3>::v, P4 = ModLimb<F, 4>::v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // eight 30-bit rounds
        const u32 q = (0u - (u32)acc[i]) & LIMB_MASK;
        // column i becomes a multiple of 2^30: acc[i] + q = ((acc[i] + 2^30 - 1) >> 30) << 30
        acc[i + 1] += ((acc[i] + LIMB_MASK) >> 30) + (u64)q * P1;
        acc[i + 2] += (u64)q * P2;
        acc[i + 3] += (u64)q * P3;
        acc[i + 4] += (u64)q * P4;
        acc[i + 8] += (u64)q << 14;
    }
    {  // one 16-bit round: 256 = 8 * 30 + 16
        const u32 q = (0u - (u32)acc[8]) & 0xffffu;
        acc[8] += q;
        acc[9] += (u64)q * P1;
        acc[10] += (u64)q * P2;
        acc[11] += (u64)q * P3;
        acc[12] += (u64)q * P4;
        acc[16] += (u64)q << 14;
    }
    // normalise columns 8..17 (low 16 bits of column 8 are zero) and realign by 16 bits
    u32 n[10];
    u64 c = 0;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        c += acc[8 + k];
        n[k] = (u32)c & LIMB_MASK;
        c >>= 30;
    }
    Fe<F> r;
#pragma unroll
    for (int k = 0; k < NLIMBS; ++k) r.l[k] = ((n[k] >> 16) | (n[k + 1] << 14)) & LIMB_MASK;
    fe_cond_sub(r);  // (T + Q m) / 2^256 < 2m
    return r;
}

template <class F> TRH_HD Fe<F> fe_mul(const Fe<F>& a, const Fe<F>& b) {
    u64 acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i)
#pragma unroll
        for (int j = 0; j < NLIMBS; ++j) acc[i + j] += (u64)a.l[i] * b.l[j];
    return fe_mont_reduce<F>(acc);
}

template <class F> TRH_HD Fe<F> fe_sqr(const Fe<F>& a) {
    u64 acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) {
        acc[2 * i] += (u64)a.l[i] * a.l[i];
        const u32 a2 = a.l[i] << 1;  // < 2^31: 4 doubled pairs + 1 square per column stay < 9 * 2^60
#pragma unroll
        for (int j = i + 1; j < NLIMBS; ++j) acc[i + j] += (u64)a2 * a.l[j];
    }
    return fe_mont_reduce<F>(acc);
}

// Montgomery -> canonical (pasta `to_repr()` as limbs): multiply by 1
template <class F> TRH_HD Fe<F> fe_from_mont(const Fe<F>& a) {
    u64 acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = k < NLIMBS ? a.l[k] : 0;
    return fe_mont_reduce<F>(acc);
}
template <class F> TRH_HD Fe<F> fe_to_mont(const Fe<F>& a) { return fe_mul(a, fe_r2<F>()); }

// a^e, e given as 32-bit words (variable time; host-side use and table setup)
template <class F> TRH_HD Fe<F> fe_pow(const Fe<F>& a, const u32* e, int nbits) {
    Fe<F> r = fe_one<F>();
    for (int i = nbits - 1; i >= 0; --i) {
        r = fe_sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1u) r = fe_mul(r, a);
    }
    return r;
}
// a^(m-2); inv(0) = 0
template <class F> TRH_HD Fe<F> fe_inv(const Fe<F>& a) {
    u32 e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = F::MOD[i];
    e[0] = 0xffffffffu;  // m - 2: word0 = 1 - 2 borrows from word1
    e[1] = F::MOD[1] - 1u;
    return fe_pow(a, e, 255);
}

// =========================================================================================
// Signed lazy domain with 29-bit limbs (the MSM's bucket arithmetic, curve.h "XYZZz").
//
// Fy holds a residue in Montgomery form with R'' = 2^261 = (2^29)^9 as nine SIGNED 32-bit limbs.
//   normalised (N): l[0..7] in [0, 2^29), l[8] signed and small; the VALUE may be negative and is only bounded (|v| < 16 m)
//   lazy (L1):      the limb-wise sum or difference of two normalised values, no carry propagation: |l[k]| < 2^30
// Why 29 bits: a column of the schoolbook product of a normalised and a lazy operand is < 9 * 2^59, and the column of TWO
// normalised products < 18 * 2^58 -- both below 2^63 with the reduction terms on top, so
//   * additions / subtractions that only feed a multiplication need no carry chain at all (fy_sub_lazy), and
//   * a * b + c * d shares ONE Montgomery reduction (fy_mul2: y3 = R (Q - x3) - Y PPP of the mixed addition).
// A reduction is more than half of a multiplication (45 of 126 multiply-adds and all of the 64-bit carry work), which is what
// this buys over the unsigned 30-bit lazy domain of rounds 1 - 2 (R' = 2^270; removed in round 6 with the NTT passes that used it).
// fy_mul(a, b) = a b / 2^261 (mod m) in (-|a b| / 2^261 - m, |a b| / 2^261]: |a|, |b| < 16 m gives (-3 m, 2 m).
// =========================================================================================
constexpr u32 YBITS = 29;
constexpr i32 YMASK = (1 << YBITS) - 1;
typedef int64_t i64;

constexpr u32 limb29_of(const u32 (&w)[8], int k) {
    const int bit = 29 * k, word = bit >> 5, sh = bit & 31;
    u64 v = w[word];
    if (word + 1 < 8) v |= (u64)w[word + 1] << 32;
    return (u32)(v >> sh) & (u32)YMASK;
}
template <class F, int K> struct YModLimb { static constexpr i32 v = (i32)limb29_of(F::MOD, K); };
template <class F, int K> struct YToLazyLimb { static constexpr i32 v = (i32)limb29_of(F::TO_LAZY29, K); };
template <class F, int K> struct YOneLimb { static constexpr i32 v = (i32)limb29_of(F::LAZY29_ONE, K); };
template <class F, int K> struct YMontOneLimb { static constexpr i32 v = (i32)limb29_of(F::ONE, K); };
static_assert(YModLimb<FpParams, 0>::v == 1 && YModLimb<FpParams, 5>::v == 0 && YModLimb<FpParams, 6>::v == 0 && YModLimb<FpParams, 7>::v == 0 &&
              YModLimb<FpParams, 8>::v == (1 << 22), "Fp modulus shape (radix 2^29)");
static_assert(YModLimb<FqParams, 0>::v == 1 && YModLimb<FqParams, 5>::v == 0 && YModLimb<FqParams, 6>::v == 0 && YModLimb<FqParams, 7>::v == 0 &&
              YModLimb<FqParams, 8>::v == (1 << 22), "Fq modulus shape (radix 2^29)");
template <class F> TRH_HD constexpr i32 ymod_limb(int k) {
    return k == 0 ? YModLimb<F, 0>::v : k == 1 ? YModLimb<F, 1>::v : k == 2 ? YModLimb<F, 2>::v : k == 3 ? YModLimb<F, 3>::v :
           k == 4 ? YModLimb<F, 4>::v : k == 5 ? YModLimb<F, 5>::v : k == 6 ? YModLimb<F, 6>::v : k == 7 ? YModLimb<F, 7>::v : YModLimb<F, 8>::v;
}

template <class F>
struct Fy {
    i32 l[NLIMBS];
};

template <class F> TRH_HD Fy<F> fy_zero() {
    Fy<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = 0;
    return r;
}
template <class F> TRH_HD Fy<F> fy_one() {
    Fy<F> r;
    r.l[0] = YOneLimb<F, 0>::v; r.l[1] = YOneLimb<F, 1>::v; r.l[2] = YOneLimb<F, 2>::v; r.l[3] = YOneLimb<F, 3>::v; r.l[4] = YOneLimb<F, 4>::v;
    r.l[5] = YOneLimb<F, 5>::v; r.l[6] = YOneLimb<F, 6>::v; r.l[7] = YOneLimb<F, 7>::v; r.l[8] = YOneLimb<F, 8>::v;
    return r;
}
template <class F> TRH_HD bool fy_is_exact_zero(const Fy<F>& a) {
    i32 o = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) o |= a.l[i];
    return o == 0;
}

// 2^22 in a register the compiler cannot see through: `q * two22 + acc` stays ONE multiply-add (the top modulus limb is 2^22)
TRH_HD i32 opaque_two22(bool negative = false) {
    i32 v = negative ? -(1 << 22) : 1 << 22;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// acc (+)= a * b on signed 32-bit operands as ONE v_mad_i64_i32.  Left to itself the compiler multiplies an operand it knows to be
// non-negative with v_mad_u64_u32 and repairs the sign of the other one with a second multiply-add on the high word plus two moves
// (seen in the NTT pass: 675 v_mov and 30 % more multiply-adds than the source has products).
TRH_HD i64 fy_prod(i32 a, i32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    i64 d;  // the carry-out operand goes to vcc (an allocated SGPR pair makes the compiler put an s_nop behind every statement)
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
    return d;
#else
    return (i64)a * b;
#endif
}
TRH_HD void fy_mac(i64& acc, i32 a, i32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
    acc += (i64)a * b;
#endif
}
// One ROW of the schoolbook product, c[k] (+)= a * b[k] for k = 0 .. 8, as one block of nine independent multiply-adds.  Single
// statements would each be followed by an s_nop (the compiler's hazard recogniser is conservative around inline assembly on
// gfx950: 1338 of them in an NTT pass); a block pays one.  INIT 0: all nine accumulate; 1: the last column starts from zero (rows
// 1 .. 8 of a product: column i + 8 is first reached there); 2: all nine start from zero (row 0).
template <int INIT> TRH_HD void fy_row(i64 (&acc)[18], int base, i32 a, const i32 (&b)[NLIMBS]) {
#if defined(__HIP_DEVICE_COMPILE__)
    i64 &c0 = acc[base], &c1 = acc[base + 1], &c2 = acc[base + 2], &c3 = acc[base + 3], &c4 = acc[base + 4], &c5 = acc[base + 5], &c6 = acc[base + 6], &c7 = acc[base + 7],
        &c8 = acc[base + 8];
    if constexpr (INIT == 2) {
        asm("v_mad_i64_i32 %0, vcc, %9, %10, 0\n\tv_mad_i64_i32 %1, vcc, %9, %11, 0\n\tv_mad_i64_i32 %2, vcc, %9, %12, 0\n\t"
            "v_mad_i64_i32 %3, vcc, %9, %13, 0\n\tv_mad_i64_i32 %4, vcc, %9, %14, 0\n\tv_mad_i64_i32 %5, vcc, %9, %15, 0\n\t"
            "v_mad_i64_i32 %6, vcc, %9, %16, 0\n\tv_mad_i64_i32 %7, vcc, %9, %17, 0\n\tv_mad_i64_i32 %8, vcc, %9, %18, 0"
            : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4), "=&v"(c5), "=&v"(c6), "=&v"(c7), "=&v"(c8)
            : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8])
            : "vcc");
    } else if constexpr (INIT == 1) {
        asm("v_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %1, vcc, %9, %11, %1\n\tv_mad_i64_i32 %2, vcc, %9, %12, %2\n\t"
            "v_mad_i64_i32 %3, vcc, %9, %13, %3\n\tv_mad_i64_i32 %4, vcc, %9, %14, %4\n\tv_mad_i64_i32 %5, vcc, %9, %15, %5\n\t"
            "v_mad_i64_i32 %6, vcc, %9, %16, %6\n\tv_mad_i64_i32 %7, vcc, %9, %17, %7\n\tv_mad_i64_i32 %8, vcc, %9, %18, 0"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "=&v"(c8)
            : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8])
            : "vcc");
    } else {
        asm("v_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %1, vcc, %9, %11, %1\n\tv_mad_i64_i32 %2, vcc, %9, %12, %2\n\t"
            "v_mad_i64_i32 %3, vcc, %9, %13, %3\n\tv_mad_i64_i32 %4, vcc, %9, %14, %4\n\tv_mad_i64_i32 %5, vcc, %9, %15, %5\n\t"
            "v_mad_i64_i32 %6, vcc, %9, %16, %6\n\tv_mad_i64_i32 %7, vcc, %9, %17, %7\n\tv_mad_i64_i32 %8, vcc, %9, %18, %8"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(c8)
            : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8])
            : "vcc");
    }
#else
    for (int k = 0; k < NLIMBS; ++k) {
        const i64 p = (i64)a * b[k];
        if (INIT == 2 || (INIT == 1 && k == NLIMBS - 1)) acc[base + k] = p; else acc[base + k] += p;
    }
#endif
}
// c[9 + k] += m * s[k]: a subtrahend (m = -1, -2) enters the result columns as one more row of multiply-adds -- one instruction
// per limb, where sign-extending s[k] and a 64-bit subtraction in the final carry chain took three
template <int M> TRH_HD void fy_row_hi(i64 (&acc)[18], const i32 (&b)[NLIMBS]) {
    static_assert(M == -1 || M == -2, "inline constants of the instruction");
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (M == -1) {
        asm("v_mad_i64_i32 %0, vcc, %9, -1, %0\n\t"
            "v_mad_i64_i32 %1, vcc, %10, -1, %1\n\t"
            "v_mad_i64_i32 %2, vcc, %11, -1, %2\n\t"
            "v_mad_i64_i32 %3, vcc, %12, -1, %3\n\t"
            "v_mad_i64_i32 %4, vcc, %13, -1, %4\n\t"
            "v_mad_i64_i32 %5, vcc, %14, -1, %5\n\t"
            "v_mad_i64_i32 %6, vcc, %15, -1, %6\n\t"
            "v_mad_i64_i32 %7, vcc, %16, -1, %7\n\t"
            "v_mad_i64_i32 %8, vcc, %17, -1, %8"
            : "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]), "+v"(acc[16]), "+v"(acc[17])
            : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8])
            : "vcc");
    } else {
        asm("v_mad_i64_i32 %0, vcc, %9, -2, %0\n\t"
            "v_mad_i64_i32 %1, vcc, %10, -2, %1\n\t"
            "v_mad_i64_i32 %2, vcc, %11, -2, %2\n\t"
            "v_mad_i64_i32 %3, vcc, %12, -2, %3\n\t"
            "v_mad_i64_i32 %4, vcc, %13, -2, %4\n\t"
            "v_mad_i64_i32 %5, vcc, %14, -2, %5\n\t"
            "v_mad_i64_i32 %6, vcc, %15, -2, %6\n\t"
            "v_mad_i64_i32 %7, vcc, %16, -2, %7\n\t"
            "v_mad_i64_i32 %8, vcc, %17, -2, %8"
            : "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]), "+v"(acc[16]), "+v"(acc[17])
            : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8])
            : "vcc");
    }
#else
    for (int k = 0; k < NLIMBS; ++k) acc[9 + k] += (i64)M * b[k];
#endif
}
// one reduction round: the five columns q touches
TRH_HD void fy_round(i64& c1, i64& c2, i64& c3, i64& c4, i64& c8, i32 q, i32 p1, i32 p2, i32 p3, i32 p4, i32 p8) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %1, vcc, %5, %7, %1\n\tv_mad_i64_i32 %2, vcc, %5, %8, %2\n\t"
        "v_mad_i64_i32 %3, vcc, %5, %9, %3\n\tv_mad_i64_i32 %4, vcc, %5, %10, %4"
        : "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c8)
        : "v"(q), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p8)
        : "vcc");
#else
    c1 += (i64)q * p1; c2 += (i64)q * p2; c3 += (i64)q * p3; c4 += (i64)q * p4; c8 += (i64)q * p8;
#endif
}
// the 9 x 9 products of a * b into the 17 columns they touch (column 17 only ever holds carries of the reduction: zeroed here)
template <class F> TRH_HD void fy_products(i64 (&acc)[18], const Fy<F>& a, const Fy<F>& b) {
    fy_row<2>(acc, 0, a.l[0], b.l);
#pragma unroll
    for (int i = 1; i < NLIMBS; ++i) fy_row<1>(acc, i, a.l[i], b.l);
    acc[17] = 0;
}
template <class F> TRH_HD void fy_products_add(i64 (&acc)[18], const Fy<F>& a, const Fy<F>& b) {
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) fy_row<0>(acc, i, a.l[i], b.l);
}
// c[k] (+)= a * b[k] for k < L as one block (the rows of a square: L = 9 .. 1).  ALL: every column starts from zero (row 0); otherwise the
// first L - 1 accumulate and the last one -- the first entry of its column -- starts from zero
template <int L, bool ALL> TRH_HD void fy_short_row(i64* c, i32 a, const i32* b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (L == 9 && ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], 0\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], 0\n\t"
            "v_mad_i64_i32 %[c2], vcc, %[a], %[b2], 0\n\t"
            "v_mad_i64_i32 %[c3], vcc, %[a], %[b3], 0\n\t"
            "v_mad_i64_i32 %[c4], vcc, %[a], %[b4], 0\n\t"
            "v_mad_i64_i32 %[c5], vcc, %[a], %[b5], 0\n\t"
            "v_mad_i64_i32 %[c6], vcc, %[a], %[b6], 0\n\t"
            "v_mad_i64_i32 %[c7], vcc, %[a], %[b7], 0\n\t"
            "v_mad_i64_i32 %[c8], vcc, %[a], %[b8], 0"
            : [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]), [c6] "=&v"(c[6]), [c7] "=&v"(c[7]), [c8] "=&v"(c[8])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7]), [b8] "v"(b[8])
            : "vcc");
    }
    else if constexpr (L == 9 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], %[c0]\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], %[c1]\n\t"
            "v_mad_i64_i32 %[c2], vcc, %[a], %[b2], %[c2]\n\t"
            "v_mad_i64_i32 %[c3], vcc, %[a], %[b3], %[c3]\n\t"
            "v_mad_i64_i32 %[c4], vcc, %[a], %[b4], %[c4]\n\t"
            "v_mad_i64_i32 %[c5], vcc, %[a], %[b5], %[c5]\n\t"
            "v_mad_i64_i32 %[c6], vcc, %[a], %[b6], %[c6]\n\t"
            "v_mad_i64_i32 %[c7], vcc, %[a], %[b7], %[c7]\n\t"
            "v_mad_i64_i32 %[c8], vcc, %[a], %[b8], 0"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [c8] "=&v"(c[8])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7]), [b8] "v"(b[8])
            : "vcc");
    }
    else if constexpr (L == 8 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], %[c0]\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], %[c1]\n\t"
            "v_mad_i64_i32 %[c2], vcc, %[a], %[b2], %[c2]\n\t"
            "v_mad_i64_i32 %[c3], vcc, %[a], %[b3], %[c3]\n\t"
            "v_mad_i64_i32 %[c4], vcc, %[a], %[b4], %[c4]\n\t"
            "v_mad_i64_i32 %[c5], vcc, %[a], %[b5], %[c5]\n\t"
            "v_mad_i64_i32 %[c6], vcc, %[a], %[b6], %[c6]\n\t"
            "v_mad_i64_i32 %[c7], vcc, %[a], %[b7], 0"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "=&v"(c[7])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7])
            : "vcc");
    }
    else if constexpr (L == 7 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], %[c0]\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], %[c1]\n\t"
            "v_mad_i64_i32 %[c2], vcc, %[a], %[b2], %[c2]\n\t"
            "v_mad_i64_i32 %[c3], vcc, %[a], %[b3], %[c3]\n\t"
            "v_mad_i64_i32 %[c4], vcc, %[a], %[b4], %[c4]\n\t"
            "v_mad_i64_i32 %[c5], vcc, %[a], %[b5], %[c5]\n\t"
            "v_mad_i64_i32 %[c6], vcc, %[a], %[b6], 0"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "=&v"(c[6])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6])
            : "vcc");
    }
    else if constexpr (L == 6 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], %[c0]\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], %[c1]\n\t"
            "v_mad_i64_i32 %[c2], vcc, %[a], %[b2], %[c2]\n\t"
            "v_mad_i64_i32 %[c3], vcc, %[a], %[b3], %[c3]\n\t"
            "v_mad_i64_i32 %[c4], vcc, %[a], %[b4], %[c4]\n\t"
            "v_mad_i64_i32 %[c5], vcc, %[a], %[b5], 0"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "=&v"(c[5])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5])
            : "vcc");
    }
    else if constexpr (L == 5 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], %[c0]\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], %[c1]\n\t"
            "v_mad_i64_i32 %[c2], vcc, %[a], %[b2], %[c2]\n\t"
            "v_mad_i64_i32 %[c3], vcc, %[a], %[b3], %[c3]\n\t"
            "v_mad_i64_i32 %[c4], vcc, %[a], %[b4], 0"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "=&v"(c[4])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4])
            : "vcc");
    }
    else if constexpr (L == 4 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], %[c0]\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], %[c1]\n\t"
            "v_mad_i64_i32 %[c2], vcc, %[a], %[b2], %[c2]\n\t"
            "v_mad_i64_i32 %[c3], vcc, %[a], %[b3], 0"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "=&v"(c[3])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3])
            : "vcc");
    }
    else if constexpr (L == 3 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], %[c0]\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], %[c1]\n\t"
            "v_mad_i64_i32 %[c2], vcc, %[a], %[b2], 0"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "=&v"(c[2])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2])
            : "vcc");
    }
    else if constexpr (L == 2 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], %[c0]\n\t"
            "v_mad_i64_i32 %[c1], vcc, %[a], %[b1], 0"
            : [c0] "+v"(c[0]), [c1] "=&v"(c[1])
            : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1])
            : "vcc");
    }
    else if constexpr (L == 1 && !ALL) {
        asm("v_mad_i64_i32 %[c0], vcc, %[a], %[b0], 0"
            : [c0] "=&v"(c[0])
            : [a] "v"(a), [b0] "v"(b[0])
            : "vcc");
    }
#else
    for (int k = 0; k < L; ++k) {
        const i64 p = (i64)a * b[k];
        if (ALL || k == L - 1) c[k] = p; else c[k] += p;
    }
#endif
}
// a^2: row i is a_i * (a_i, 2 a_{i+1}, ..., 2 a_8) into columns 2 i .. i + 8
template <class F> TRH_HD void fy_squares(i64 (&acc)[18], const Fy<F>& a) {
    i32 d[NLIMBS][NLIMBS];  // d[i] = (a_i, 2 a_{i+1}, ...): the doubled limbs are shared (< 2^30 in magnitude: a is normalised)
    i32 a2[NLIMBS];
#pragma unroll
    for (int j = 0; j < NLIMBS; ++j) a2[j] = a.l[j] * 2;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) {
        d[i][0] = a.l[i];
#pragma unroll
        for (int j = i + 1; j < NLIMBS; ++j) d[i][j - i] = a2[j];
    }
    fy_short_row<9, true>(&acc[0], a.l[0], d[0]);
    fy_short_row<8, false>(&acc[2], a.l[1], d[1]);
    fy_short_row<7, false>(&acc[4], a.l[2], d[2]);
    fy_short_row<6, false>(&acc[6], a.l[3], d[3]);
    fy_short_row<5, false>(&acc[8], a.l[4], d[4]);
    fy_short_row<4, false>(&acc[10], a.l[5], d[5]);
    fy_short_row<3, false>(&acc[12], a.l[6], d[6]);
    fy_short_row<2, false>(&acc[14], a.l[7], d[7]);
    fy_short_row<1, false>(&acc[16], a.l[8], d[8]);
    acc[17] = 0;
}

// nine uniform 29-bit rounds on signed columns; result = value / 2^261 (mod m) - s1 - 2 s2, normalised.  The subtrahends ride in the
// final carry chain (the difference that follows a product would otherwise be a second chain): SUB 0 none, 1 s1, 2 s1 and 2 s2.
// A round removes r = column mod 2^29 by adding -r m (m = 1 mod 2^29): the low limb of -r m cancels r, so the carry into the next
// column is simply floor(column / 2^29) and the other limbs of m enter as r * (-m_k) -- three instructions (and, shift, add) next to the
// five multiply-adds.  The result lies in (-v / 2^261 - m, v / 2^261]: fine for the signed domain.  NONNEG picks the mirror image
// (q = -r mod 2^29, + q m, carry = ceil): result in [v / 2^261, v / 2^261 + m), i.e. non-negative for v >= 0 -- needed where the
// result is stored as eight words (fy_store) -- at five instructions per round.
template <class F, int SUB, bool NONNEG = false> TRH_HD Fy<F> fy_reduce_sub(i64 (&acc)[18], const Fy<F>* s1, const Fy<F>* s2) {
    constexpr i32 P1 = YModLimb<F, 1>::v, P2 = YModLimb<F, 2>::v, P3 = YModLimb<F, 3>::v, P4 = YModLimb<F, 4>::v;
    const i32 two22 = opaque_two22(!NONNEG);  // +-2^22
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (NONNEG) {
            const i32 q = (i32)((0u - (u32)acc[i]) & (u32)YMASK);
            // column i + q is a multiple of 2^29: its quotient is ceil(acc[i] / 2^29) (arithmetic shift = floor)
            acc[i + 1] += (acc[i] + YMASK) >> YBITS;
            fy_round(acc[i + 1], acc[i + 2], acc[i + 3], acc[i + 4], acc[i + 8], q, P1, P2, P3, P4, two22);
        } else {
            const i32 r = (i32)((u32)acc[i] & (u32)YMASK);
            acc[i + 1] += acc[i] >> YBITS;
            fy_round(acc[i + 1], acc[i + 2], acc[i + 3], acc[i + 4], acc[i + 8], r, -P1, -P2, -P3, -P4, two22);
        }
    }
    if (SUB >= 1) fy_row_hi<-1>(acc, s1->l);
    if (SUB == 2) fy_row_hi<-2>(acc, s2->l);
    Fy<F> r;
    i64 c = 0;
#pragma unroll
    for (int k = 0; k < NLIMBS; ++k) {
        c += acc[9 + k];
        if (k < NLIMBS - 1) {
            r.l[k] = (i32)((u32)c & (u32)YMASK);
            c >>= YBITS;
        }
    }
    r.l[8] = (i32)c;  // signed top limb: |value| < 2^260
    return r;
}
template <class F> TRH_HD Fy<F> fy_reduce(i64 (&acc)[18]) { return fy_reduce_sub<F, 0>(acc, nullptr, nullptr); }
// a normalised or lazy, b normalised (or the other way round): |a_i b_j| < 2^59
template <class F> TRH_HD Fy<F> fy_mul(const Fy<F>& a, const Fy<F>& b) {
    i64 acc[18];
    fy_products(acc, a, b);
    return fy_reduce<F>(acc);
}
// the same with a non-negative result for non-negative operands (below a b / 2^261 + m): for values that are stored as words
template <class F> TRH_HD Fy<F> fy_mul_nonneg(const Fy<F>& a, const Fy<F>& b) {
    i64 acc[18];
    fy_products(acc, a, b);
    return fy_reduce_sub<F, 0, true>(acc, nullptr, nullptr);
}
// a normalised
template <class F> TRH_HD Fy<F> fy_sqr(const Fy<F>& a) {
    i64 acc[18];
    fy_squares(acc, a);
    return fy_reduce<F>(acc);
}
// a b + c d with ONE reduction.  Column bound: (a or b lazy, the other normalised) + (c, d normalised) <= 9 * 2^59 + 9 * 2^58 < 2^62.8
template <class F> TRH_HD Fy<F> fy_mul2(const Fy<F>& a, const Fy<F>& b, const Fy<F>& c, const Fy<F>& d) {
    i64 acc[18];
    fy_products(acc, a, b);
    fy_products_add(acc, c, d);
    return fy_reduce<F>(acc);
}
// a b - s (U2 - X, S2 - Y of the mixed addition): the subtraction rides in the product's carry chain
template <class F> TRH_HD Fy<F> fy_mul_sub(const Fy<F>& a, const Fy<F>& b, const Fy<F>& s) {
    i64 acc[18];
    fy_products(acc, a, b);
    return fy_reduce_sub<F, 1>(acc, &s, nullptr);
}
// a^2 - s1 - 2 s2 (x3 = R^2 - PPP - 2 Q)
template <class F> TRH_HD Fy<F> fy_sqr_sub_sub2(const Fy<F>& a, const Fy<F>& s1, const Fy<F>& s2) {
    i64 acc[18];
    fy_squares(acc, a);
    return fy_reduce_sub<F, 2>(acc, &s1, &s2);
}
// carry propagation: any limbs (|l[k]| < 2^31 - 2^3) -> normalised
template <class F> TRH_HD Fy<F> fy_norm(const Fy<F>& a) {
    Fy<F> r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS - 1; ++i) {
        const i32 v = a.l[i] + c;
        r.l[i] = v & YMASK;
        c = v >> YBITS;
    }
    r.l[8] = a.l[8] + c;
    return r;
}
// normalised -> balanced limbs: l[0..7] in [-2^28, 2^28), same value.  A table constant in this form halves the column bound of a
// product, so the OTHER operand may be any limbs that fit 32 bits (the NTT's butterflies then normalise half as often)
template <class F> TRH_HD Fy<F> fy_balance(const Fy<F>& a) {
    Fy<F> r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS - 1; ++i) {
        const i32 v = a.l[i] + c;                  // [0, 2^29]
        c = (v + (1 << (YBITS - 1))) >> YBITS;     // 1 when v >= 2^28
        r.l[i] = v - (c << YBITS);
    }
    r.l[8] = a.l[8] + c;
    return r;
}
// limb-wise, no carries: the result only feeds ONE multiplication (as its lazy operand) or a fy_norm
template <class F> TRH_HD Fy<F> fy_add_lazy(const Fy<F>& a, const Fy<F>& b) {
    Fy<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <class F> TRH_HD Fy<F> fy_sub_lazy(const Fy<F>& a, const Fy<F>& b) {
    Fy<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = a.l[i] - b.l[i];
    return r;
}
template <class F> TRH_HD Fy<F> fy_neg_lazy(const Fy<F>& a) {
    Fy<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = -a.l[i];
    return r;
}
// normalising forms (one carry chain)
template <class F> TRH_HD Fy<F> fy_add(const Fy<F>& a, const Fy<F>& b) {
    Fy<F> r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS - 1; ++i) {
        const i32 v = a.l[i] + b.l[i] + c;
        r.l[i] = v & YMASK;
        c = v >> YBITS;
    }
    r.l[8] = a.l[8] + b.l[8] + c;
    return r;
}
template <class F> TRH_HD Fy<F> fy_sub(const Fy<F>& a, const Fy<F>& b) {
    Fy<F> r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS - 1; ++i) {
        const i32 v = a.l[i] - b.l[i] + c;
        r.l[i] = v & YMASK;
        c = v >> YBITS;
    }
    r.l[8] = a.l[8] - b.l[8] + c;
    return r;
}
// a - b - 2 c, normalised (x3 = R^2 - PPP - 2 Q in one chain)
template <class F> TRH_HD Fy<F> fy_sub_sub2(const Fy<F>& a, const Fy<F>& b, const Fy<F>& c2) {
    Fy<F> r;
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS - 1; ++i) {
        const i32 v = a.l[i] - b.l[i] - 2 * c2.l[i] + c;  // > -2^31
        r.l[i] = v & YMASK;
        c = v >> YBITS;
    }
    r.l[8] = a.l[8] - b.l[8] - 2 * c2.l[8] + c;
    return r;
}
// the cheap half of the test below: false for all but 33 of the 2^29 low limbs
template <class F> TRH_HD bool fy_maybe_zero_mod(const Fy<F>& a) { return (u32)(a.l[0] + 16) % (u32)(YMASK + 1) <= 32u; }
// normalised value == 0 (mod m)?  |value| < 16 m, so it would be j m with |j| <= 16, whose low limb is j mod 2^29 (m = 1 mod 2^29)
template <class F> TRH_HD bool fy_is_zero_mod(const Fy<F>& a) {
    const i32 l0 = a.l[0];
    i32 j;
    if (l0 <= 16) j = l0;
    else if (l0 >= YMASK + 1 - 16) j = l0 - (YMASK + 1);
    else return false;
    i64 carry = 0;
    i32 diff = 0;
    for (int i = 0; i < NLIMBS - 1; ++i) {
        const i64 t = (i64)j * ymod_limb<F>(i) + carry;
        diff |= (i32)((u32)t & (u32)YMASK) ^ a.l[i];
        carry = t >> YBITS;
    }
    diff |= (i32)((i64)j * ymod_limb<F>(8) + carry) ^ a.l[8];
    return diff == 0;
}
// memory words <-> limbs; the stored value must be in [0, 2^256)
template <class F> TRH_HD Fy<F> fy_load(u32 w0, u32 w1, u32 w2, u32 w3, u32 w4, u32 w5, u32 w6, u32 w7) {
    Fy<F> r;
    const u32 M = (u32)YMASK;
    r.l[0] = (i32)(w0 & M);
    r.l[1] = (i32)(((w0 >> 29) | (w1 << 3)) & M);
    r.l[2] = (i32)(((w1 >> 26) | (w2 << 6)) & M);
    r.l[3] = (i32)(((w2 >> 23) | (w3 << 9)) & M);
    r.l[4] = (i32)(((w3 >> 20) | (w4 << 12)) & M);
    r.l[5] = (i32)(((w4 >> 17) | (w5 << 15)) & M);
    r.l[6] = (i32)(((w5 >> 14) | (w6 << 18)) & M);
    r.l[7] = (i32)(((w6 >> 11) | (w7 << 21)) & M);
    r.l[8] = (i32)(w7 >> 8);
    return r;
}
template <class F> TRH_HD void fy_store(const Fy<F>& a, u32* w) {
    const u32* l = (const u32*)a.l;
    w[0] = l[0] | (l[1] << 29);
    w[1] = (l[1] >> 3) | (l[2] << 26);
    w[2] = (l[2] >> 6) | (l[3] << 23);
    w[3] = (l[3] >> 9) | (l[4] << 20);
    w[4] = (l[4] >> 12) | (l[5] << 17);
    w[5] = (l[5] >> 15) | (l[6] << 14);
    w[6] = (l[6] >> 18) | (l[7] << 11);
    w[7] = (l[7] >> 21) | (l[8] << 8);
}
// canonical Montgomery-R element -> this domain (result in [0, m (1 + 2^-7)))
template <class F> TRH_HD Fy<F> fy_from_fe(const Fe<F>& a) {
    u32 w[8];
    fe_store(a, w);
    const Fy<F> x = fy_load<F>(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
    Fy<F> c;
    c.l[0] = YToLazyLimb<F, 0>::v; c.l[1] = YToLazyLimb<F, 1>::v; c.l[2] = YToLazyLimb<F, 2>::v; c.l[3] = YToLazyLimb<F, 3>::v; c.l[4] = YToLazyLimb<F, 4>::v;
    c.l[5] = YToLazyLimb<F, 5>::v; c.l[6] = YToLazyLimb<F, 6>::v; c.l[7] = YToLazyLimb<F, 7>::v; c.l[8] = YToLazyLimb<F, 8>::v;
    return fy_mul_nonneg(x, c);
}
// normalised (|value| < 16 m) -> canonical Montgomery-R element: multiply by 2^256 mod m here, then bring (-m/8, 9 m / 8) into [0, m)
template <class F> TRH_HD Fe<F> fy_to_fe(const Fy<F>& a) {
    Fy<F> c;
    c.l[0] = YMontOneLimb<F, 0>::v; c.l[1] = YMontOneLimb<F, 1>::v; c.l[2] = YMontOneLimb<F, 2>::v; c.l[3] = YMontOneLimb<F, 3>::v; c.l[4] = YMontOneLimb<F, 4>::v;
    c.l[5] = YMontOneLimb<F, 5>::v; c.l[6] = YMontOneLimb<F, 6>::v; c.l[7] = YMontOneLimb<F, 7>::v; c.l[8] = YMontOneLimb<F, 8>::v;
    Fy<F> t = fy_mul_nonneg(a, c);
    Fy<F> mm;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) mm.l[i] = ymod_limb<F>(i);
    t = fy_add(t, mm);  // (7 m / 8, 17 m / 8): non-negative, below 2^256
    u32 w[8];
    fy_store(t, w);
    Fe<F> r = fe_load<F>(w);
    fe_cond_sub(r);  // [0, 2m) steps
    fe_cond_sub(r);
    return r;
}

}  // namespace trh
