// Pasta field arithmetic for gfx950: 255-bit Montgomery (R = 2^256) on 8 x u32 limbs.
//
// Replaces pasta_curves 0.4.1 `Fp` / `Fq` (fields/fp.rs, fields/fq.rs; pinned at
// /root/reference/Cargo.lock:847-858, used by the reference at src/test_utils.rs:2 and
// src/circuits/tables/even_bits.rs:250-262).  Memory format is identical to the Rust one:
// four u64 little-endian limbs, Montgomery form, fully reduced -- the same bytes read as
// eight u32 limbs here, so buffers cross the C ABI without repacking.
//
// Both moduli have the shape  m = 2^254 + t,  t < 2^126,  m = 1 (mod 2^32):
//   limbs(m) = [1, M1, M2, M3, 0, 0, 0, 0x40000000]
// so in word-serial Montgomery reduction the quotient digit is  q = -T[i] mod 2^32  (no
// multiply: -m^-1 = -1 mod 2^32), q*m needs only three 32x32 multiplies (M1..M3), the low word
// is a pure carry and the top word is a shift by 30.  A field multiply is therefore
// 64 (product) + 24 (reduction) v_mad_u64_u32 instead of 64 + 72.
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

struct FpParams {  // Pallas base field = Vesta scalar field
    static constexpr u32 M1 = 0x992d30edu, M2 = 0x094cf91bu, M3 = 0x224698fcu;
    static constexpr int ID = 0;
    TRH_HD static constexpr u32 one(int i) {  // R = 2^256 mod p
        constexpr u32 v[8] = {0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
        return v[i];
    }
    TRH_HD static constexpr u32 r2(int i) {  // R^2 mod p
        constexpr u32 v[8] = {0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu, 0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu};
        return v[i];
    }
};
struct FqParams {  // Vesta base field = Pallas scalar field
    static constexpr u32 M1 = 0x8c46eb21u, M2 = 0x0994a8ddu, M3 = 0x224698fcu;
    static constexpr int ID = 1;
    TRH_HD static constexpr u32 one(int i) {
        constexpr u32 v[8] = {0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
        return v[i];
    }
    TRH_HD static constexpr u32 r2(int i) {
        constexpr u32 v[8] = {0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu};
        return v[i];
    }
};

template <class F>
TRH_HD constexpr u32 mod_limb(int i) {
    return i == 0 ? 1u : i == 1 ? F::M1 : i == 2 ? F::M2 : i == 3 ? F::M3 : i == 7 ? 0x40000000u : 0u;
}

template <class F>
struct Fe {
    u32 l[8];
};

template <class F> TRH_HD Fe<F> fe_zero() {
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = 0;
    return r;
}
template <class F> TRH_HD Fe<F> fe_one() {
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = F::one(i);
    return r;
}
template <class F> TRH_HD Fe<F> fe_r2() {
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = F::r2(i);
    return r;
}
template <class F> TRH_HD bool fe_is_zero(const Fe<F>& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.l[i];
    return o == 0;
}
template <class F> TRH_HD bool fe_eq(const Fe<F>& a, const Fe<F>& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// r = a - m if a >= m else a   (a < 2m)
template <class F> TRH_HD void fe_cond_sub(Fe<F>& a) {
    u32 t[8];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = __builtin_subc(a.l[i], mod_limb<F>(i), borrow, &borrow);
#pragma unroll
    for (int i = 0; i < 8; ++i) a.l[i] = borrow ? a.l[i] : t[i];
}

template <class F> TRH_HD Fe<F> fe_add(const Fe<F>& a, const Fe<F>& b) {
    Fe<F> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    fe_cond_sub(r);  // a + b < 2m < 2^256: no carry out
    return r;
}
template <class F> TRH_HD Fe<F> fe_sub(const Fe<F>& a, const Fe<F>& b) {
    Fe<F> r;
    u32 bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = __builtin_subc(a.l[i], b.l[i], bw, &bw);
    u32 mask = 0u - bw, c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = __builtin_addc(r.l[i], mod_limb<F>(i) & mask, c, &c);
    return r;
}
template <class F> TRH_HD Fe<F> fe_neg(const Fe<F>& a) { return fe_sub(fe_zero<F>(), a); }
template <class F> TRH_HD Fe<F> fe_dbl(const Fe<F>& a) { return fe_add(a, a); }

// Word-serial Montgomery reduction of a 16-word value T < m * 2^256; returns T / 2^256 mod m.
template <class F> TRH_HD Fe<F> fe_mont_reduce(u32 (&t)[16]) {
    u32 hi = 0;  // deferred carry into word i+9
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 q = 0u - t[i];
        u64 uv;
        u32 c = (t[i] != 0) ? 1u : 0u;  // t[i] + q*1 = 2^32 or 0
        uv = (u64)q * F::M1 + t[i + 1] + c; t[i + 1] = (u32)uv; c = (u32)(uv >> 32);
        uv = (u64)q * F::M2 + t[i + 2] + c; t[i + 2] = (u32)uv; c = (u32)(uv >> 32);
        uv = (u64)q * F::M3 + t[i + 3] + c; t[i + 3] = (u32)uv; c = (u32)(uv >> 32);
        t[i + 4] = __builtin_addc(t[i + 4], 0u, c, &c);
        t[i + 5] = __builtin_addc(t[i + 5], 0u, c, &c);
        t[i + 6] = __builtin_addc(t[i + 6], 0u, c, &c);
        t[i + 7] = __builtin_addc(t[i + 7], q << 30, c, &c);
        uv = (u64)t[i + 8] + (q >> 2) + c + hi;
        t[i + 8] = (u32)uv;
        hi = (u32)(uv >> 32);
    }
    // hi == 0 here: (T + Q*m) / 2^256 < 2m < 2^256
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = t[i + 8];
    fe_cond_sub(r);
    return r;
}

template <class F> TRH_HD Fe<F> fe_mul(const Fe<F>& a, const Fe<F>& b) {
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u64 uv = (u64)a.l[i] * b.l[j] + (i ? t[i + j] : 0u) + c;
            t[i + j] = (u32)uv;
            c = (u32)(uv >> 32);
        }
        t[i + 8] = c;
    }
    return fe_mont_reduce<F>(t);
}

template <class F> TRH_HD Fe<F> fe_sqr(const Fe<F>& a) {
    // off-diagonal products once, doubled, plus the diagonal: 36 multiplies instead of 64
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        u32 c = 0;
#pragma unroll
        for (int j = i + 1; j < 8; ++j) {
            u64 uv = (u64)a.l[i] * a.l[j] + t[i + j] + c;
            t[i + j] = (u32)uv;
            c = (u32)(uv >> 32);
        }
        t[i + 8] = c;
    }
    // double
    u32 top = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        u32 nt = t[i] >> 31;
        t[i] = (t[i] << 1) | top;
        top = nt;
    }
    // add diagonal
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 d = (u64)a.l[i] * a.l[i];
        t[2 * i] = __builtin_addc(t[2 * i], (u32)d, c, &c);
        t[2 * i + 1] = __builtin_addc(t[2 * i + 1], (u32)(d >> 32), c, &c);
    }
    return fe_mont_reduce<F>(t);
}

// Montgomery -> canonical (pasta `to_repr()` as limbs): multiply by 1
template <class F> TRH_HD Fe<F> fe_from_mont(const Fe<F>& a) {
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { t[i] = a.l[i]; t[i + 8] = 0; }
    return fe_mont_reduce<F>(t);
}
template <class F> TRH_HD Fe<F> fe_to_mont(const Fe<F>& a) { return fe_mul(a, fe_r2<F>()); }

// a^e, e given as 8 u32 limbs (variable time; host-side use and table setup)
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
    for (int i = 0; i < 8; ++i) e[i] = mod_limb<F>(i);
    e[0] = 0xffffffffu;  // m - 2: limb0 = 1 - 2 borrows from limb1
    e[1] = mod_limb<F>(1) - 1u;
    return fe_pow(a, e, 255);
}

}  // namespace trh
