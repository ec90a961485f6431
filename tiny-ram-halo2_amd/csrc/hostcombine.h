// Host side of the MSM: Horner over the per-window sums (acc = 2^c acc + S_j, c doublings per window) and the affine
// normalisation of the result -- the tail of halo2_proofs' `multiexp_serial` (arithmetic.rs; crate pinned at
// /root/reference/Cargo.lock:619-621) that libtrh leaves on the CPU because it is a serial chain of ~255 doublings.
// The device code keeps field elements as nine 30-bit limbs (field.h), which is the right shape for v_mad_u64_u32 but
// slow on a CPU (81 multiplies per product): here the same Montgomery values (R = 2^256, memory format = 4 x u64) are
// multiplied with 64 x 64 -> 128-bit products (16 + 16 multiplies), 4x faster.  Plain C++, no HIP: unit-tested on the
// CPU against the generic implementation (tests/test_hostcombine.py).
#pragma once
#include <stdint.h>
#include <string.h>

namespace trh {
namespace hostcombine {

typedef unsigned __int128 u128;
struct H { uint64_t l[4]; };

template <class F> struct Consts {
    H m, one;      // modulus, R mod m
    uint64_t inv;  // -m^-1 mod 2^64
    Consts() {
        for (int i = 0; i < 4; ++i) m.l[i] = (uint64_t)F::MOD[2 * i] | ((uint64_t)F::MOD[2 * i + 1] << 32);
        uint64_t x = 1;  // Newton: x <- x (2 - m0 x) doubles the number of correct low bits
        for (int i = 0; i < 6; ++i) x *= 2 - m.l[0] * x;
        inv = 0 - x;
        H z = {{0, 0, 0, 0}};
        one = sub_raw(sub_raw(sub_raw(z, m), m), m);  // 2^256 - 3m: both moduli sit just above 2^254
    }
    static H sub_raw(const H& a, const H& b) {
        H r; u128 br = 0;
        for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
        return r;
    }
};
template <class F> inline const Consts<F>& consts() { static const Consts<F> c; return c; }

inline bool is_zero(const H& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline bool eq(const H& a, const H& b) { return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0; }
inline bool geq(const H& a, const H& b) {
    for (int i = 3; i >= 0; --i) if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    return true;
}
template <class F> inline H add(const H& a, const H& b) {
    H r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    return geq(r, consts<F>().m) ? Consts<F>::sub_raw(r, consts<F>().m) : r;  // a + b < 2m < 2^256
}
inline H add_raw(const H& a, const H& b) {
    H r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    return r;
}
template <class F> inline H sub(const H& a, const H& b) {
    if (geq(a, b)) return Consts<F>::sub_raw(a, b);
    return Consts<F>::sub_raw(add_raw(a, consts<F>().m), b);  // a + m < 2^256
}
template <class F> inline H dbl(const H& a) { return add<F>(a, a); }
// Montgomery product a b / 2^256 mod m (CIOS, unrolled).  Both moduli are 2^254 + a 126-bit number: limbs (m0, m1, 0, 2^62), so a reduction
// round is two multiplications and a shift instead of four multiplications.
#define TRH_HC_REDUCE_ROUND                                                       \
    {                                                                             \
        const uint64_t q = t0 * inv;                                              \
        u128 c = ((u128)q * m0 + t0) >> 64;                                       \
        c += (u128)q * m1 + t1; t0 = (uint64_t)c; c >>= 64;                       \
        c += t2; t1 = (uint64_t)c; c >>= 64;                                      \
        c += ((u128)q << 62) + t3; t2 = (uint64_t)c; c >>= 64;                    \
        c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);                 \
    }
template <class F> inline H mul(const H& a, const H& b) {
    const H& m = consts<F>().m;
    const uint64_t inv = consts<F>().inv, m0 = m.l[0], m1 = m.l[1];
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
#define TRH_HC_ROUND(ai)                                                          \
    {                                                                             \
        u128 c = (u128)(ai) * b.l[0] + t0; t0 = (uint64_t)c; c >>= 64;            \
        c += (u128)(ai) * b.l[1] + t1; t1 = (uint64_t)c; c >>= 64;                \
        c += (u128)(ai) * b.l[2] + t2; t2 = (uint64_t)c; c >>= 64;                \
        c += (u128)(ai) * b.l[3] + t3; t3 = (uint64_t)c; c >>= 64;                \
        c += t4; t4 = (uint64_t)c; t5 = (uint64_t)(c >> 64);                      \
    }                                                                             \
    TRH_HC_REDUCE_ROUND
    TRH_HC_ROUND(a.l[0]) TRH_HC_ROUND(a.l[1]) TRH_HC_ROUND(a.l[2]) TRH_HC_ROUND(a.l[3])
#undef TRH_HC_ROUND
    H r = {{t0, t1, t2, t3}};
    return (t4 || geq(r, m)) ? Consts<F>::sub_raw(r, m) : r;
}
// a^2 / 2^256 mod m: the six cross products once, doubled, plus the four squares (10 multiplications for the 8-limb square), then four reduction rounds
template <class F> inline H sqr(const H& a) {
    const H& m = consts<F>().m;
    const uint64_t inv = consts<F>().inv, m0 = m.l[0], m1 = m.l[1];
    uint64_t w[8];
    {
        u128 c = (u128)a.l[0] * a.l[1]; w[1] = (uint64_t)c; c >>= 64;
        c += (u128)a.l[0] * a.l[2]; w[2] = (uint64_t)c; c >>= 64;
        c += (u128)a.l[0] * a.l[3]; w[3] = (uint64_t)c; w[4] = (uint64_t)(c >> 64);
        c = (u128)a.l[1] * a.l[2] + w[3]; w[3] = (uint64_t)c; c >>= 64;
        c += (u128)a.l[1] * a.l[3] + w[4]; w[4] = (uint64_t)c; w[5] = (uint64_t)(c >> 64);
        c = (u128)a.l[2] * a.l[3] + w[5]; w[5] = (uint64_t)c; w[6] = (uint64_t)(c >> 64);
        w[7] = w[6] >> 63; w[6] = (w[6] << 1) | (w[5] >> 63); w[5] = (w[5] << 1) | (w[4] >> 63); w[4] = (w[4] << 1) | (w[3] >> 63);
        w[3] = (w[3] << 1) | (w[2] >> 63); w[2] = (w[2] << 1) | (w[1] >> 63); w[1] <<= 1;
        c = (u128)a.l[0] * a.l[0]; w[0] = (uint64_t)c; c >>= 64;
        c += w[1]; w[1] = (uint64_t)c; c >>= 64;
        c += (u128)a.l[1] * a.l[1] + w[2]; w[2] = (uint64_t)c; c >>= 64;
        c += w[3]; w[3] = (uint64_t)c; c >>= 64;
        c += (u128)a.l[2] * a.l[2] + w[4]; w[4] = (uint64_t)c; c >>= 64;
        c += w[5]; w[5] = (uint64_t)c; c >>= 64;
        c += (u128)a.l[3] * a.l[3] + w[6]; w[6] = (uint64_t)c; c >>= 64;
        w[7] += (uint64_t)c;
    }
    // Montgomery reduction of the 512-bit square, a word at a time: (t0 .. t3) is the running low part, the high words enter one per round
    uint64_t t0 = w[0], t1 = w[1], t2 = w[2], t3 = w[3], t4 = 0, t5 = 0, carry = 0;
#define TRH_HC_SQ_ROUND(hi)                                                       \
    {                                                                             \
        u128 c = (u128)(hi) + carry; t4 = (uint64_t)c; t5 = (uint64_t)(c >> 64);  \
        TRH_HC_REDUCE_ROUND                                                       \
        carry = t4;                                                               \
    }
    TRH_HC_SQ_ROUND(w[4]) TRH_HC_SQ_ROUND(w[5]) TRH_HC_SQ_ROUND(w[6]) TRH_HC_SQ_ROUND(w[7])
#undef TRH_HC_SQ_ROUND
    H r = {{t0, t1, t2, t3}};
    return (carry || geq(r, m)) ? Consts<F>::sub_raw(r, m) : r;
}
#undef TRH_HC_REDUCE_ROUND
template <class F> inline H inv(const H& a) {  // a^(m - 2); inv(0) = 0
    H e = consts<F>().m;
    e.l[0] -= 2;  // the low limb ends in ...00000001: no borrow
    H r = consts<F>().one;
    for (int i = 254; i >= 0; --i) {
        r = sqr<F>(r);
        if ((e.l[i >> 6] >> (i & 63)) & 1) r = mul<F>(r, a);
    }
    return r;
}

struct P { H x, y, zz, zzz; };  // XYZZ: x = X / ZZ, y = Y / ZZZ; identity <=> zz == 0 (as curve.h)
inline bool is_identity(const P& p) { return is_zero(p.zz); }
inline P identity() { P p; memset(&p, 0, sizeof(p)); return p; }

// dbl-2008-s-1, a = 0
template <class F> inline P pdbl(const P& p) {
    if (is_identity(p)) return p;
    const H U = dbl<F>(p.y);
    if (is_zero(U)) return identity();
    const H V = sqr<F>(U), W = mul<F>(U, V), S = mul<F>(p.x, V);
    const H xx = sqr<F>(p.x), M = add<F>(dbl<F>(xx), xx);
    P r;
    r.x = sub<F>(sqr<F>(M), dbl<F>(S));
    r.y = sub<F>(mul<F>(M, sub<F>(S, r.x)), mul<F>(W, p.y));
    r.zz = mul<F>(V, p.zz);
    r.zzz = mul<F>(W, p.zzz);
    return r;
}
// add-2008-s with the exceptional cases
template <class F> inline P padd(const P& a, const P& b) {
    if (is_identity(a)) return b;
    if (is_identity(b)) return a;
    const H U1 = mul<F>(a.x, b.zz), U2 = mul<F>(b.x, a.zz), S1 = mul<F>(a.y, b.zzz), S2 = mul<F>(b.y, a.zzz);
    const H Pd = sub<F>(U2, U1), R = sub<F>(S2, S1);
    if (is_zero(Pd)) return is_zero(R) ? pdbl<F>(a) : identity();
    const H PP = sqr<F>(Pd), PPP = mul<F>(Pd, PP), Q = mul<F>(U1, PP);
    P r;
    r.x = sub<F>(sub<F>(sqr<F>(R), PPP), dbl<F>(Q));
    r.y = sub<F>(mul<F>(R, sub<F>(Q, r.x)), mul<F>(S1, PPP));
    r.zz = mul<F>(mul<F>(a.zz, b.zz), PP);
    r.zzz = mul<F>(mul<F>(a.zzz, b.zzz), PPP);
    return r;
}

// window sums (XYZZ, canonical Montgomery words: 4 x 4 u64 per point) -> normalised Jacobian (X, Y, Z = 1; identity all-zero)
template <class F> inline void combine_windows(const uint64_t* ws /* W x 16 u64 */, int W, int cb, uint64_t* out_xyz /* 12 u64 */) {
    P acc = identity();
    for (int j = W - 1; j >= 0; --j) {
        if (!is_identity(acc)) for (int k = 0; k < cb; ++k) acc = pdbl<F>(acc);
        P s;
        memcpy(&s, ws + 16 * (size_t)j, sizeof(P));
        acc = padd<F>(acc, s);
    }
    memset(out_xyz, 0, 96);
    if (is_identity(acc)) return;
    // 1 / ZZZ, then 1 / ZZ = ZZZ^-2 ZZ^2  (ZZ^3 = ZZZ^2)
    const H zzz_inv = inv<F>(acc.zzz);
    const H zz_inv = mul<F>(sqr<F>(zzz_inv), sqr<F>(acc.zz));
    const H x = mul<F>(acc.x, zz_inv), y = mul<F>(acc.y, zzz_inv);
    memcpy(out_xyz, &x, 32);
    memcpy(out_xyz + 4, &y, 32);
    memcpy(out_xyz + 8, &consts<F>().one, 32);
}

// Horner over one result's W window sums (XYZZ out, not normalised)
template <class F> inline P horner(const uint64_t* w /* W x 16 u64 */, int W, int cb) {
    P a = identity();
    for (int j = W - 1; j >= 0; --j) {
        if (!is_identity(a)) for (int k = 0; k < cb; ++k) a = pdbl<F>(a);
        P s;
        memcpy(&s, w + 16 * (size_t)j, sizeof(P));
        a = padd<F>(a, s);
    }
    return a;
}
// nb points -> normalised Jacobian with ONE field inversion (Montgomery's trick: prefix products of the ZZZ's, one inversion, two multiplications
// per item on the way back).  An inversion is 255 squarings + ~130 multiplications = ~12 us here; a batch of 64 commitments spent 0.8 ms of host
// time in them between two launches, an IPA round 24 us.
template <class F> inline void normalise_batch(const P* acc, size_t nb, uint64_t* out_xyz /* nb x 12 u64 */) {
    constexpr size_t CHUNK = 256;
    H pre[CHUNK];
    for (size_t b0 = 0; b0 < nb; b0 += CHUNK) {
        const size_t cur = nb - b0 < CHUNK ? nb - b0 : CHUNK;
        H run = consts<F>().one;
        for (size_t i = 0; i < cur; ++i) {
            pre[i] = run;                                   // product of the ZZZ's of the non-identity items before i
            if (!is_identity(acc[b0 + i])) run = mul<F>(run, acc[b0 + i].zzz);
        }
        H inv_run = inv<F>(run);                            // (run = 1 when every item is the identity)
        for (size_t i = cur; i-- > 0;) {
            const P& a = acc[b0 + i];
            uint64_t* o = out_xyz + (b0 + i) * 12;
            memset(o, 0, 96);
            if (is_identity(a)) continue;
            const H zzz_inv = mul<F>(inv_run, pre[i]);
            inv_run = mul<F>(inv_run, a.zzz);
            const H zz_inv = mul<F>(sqr<F>(zzz_inv), sqr<F>(a.zz));
            const H x = mul<F>(a.x, zz_inv), y = mul<F>(a.y, zzz_inv);
            memcpy(o, &x, 32);
            memcpy(o + 4, &y, 32);
            memcpy(o + 8, &consts<F>().one, 32);
        }
    }
}

// `batch` MSMs at once (W window sums each, back to back): the Horners one after the other, one inversion for all of them
template <class F> inline void combine_windows_batch(const uint64_t* ws /* batch x W x 16 u64 */, int W, int cb, size_t batch, uint64_t* out_xyz /* batch x 12 u64 */) {
    if (batch == 1) { combine_windows<F>(ws, W, cb, out_xyz); return; }
    constexpr size_t CHUNK = 256;
    P acc[CHUNK];
    for (size_t b0 = 0; b0 < batch; b0 += CHUNK) {
        const size_t nb = batch - b0 < CHUNK ? batch - b0 : CHUNK;
        for (size_t i = 0; i < nb; ++i) acc[i] = horner<F>(ws + (b0 + i) * (size_t)W * 16, W, cb);
        normalise_batch<F>(acc, nb, out_xyz + b0 * 12);
    }
}

}  // namespace hostcombine
}  // namespace trh
