// Pallas / Vesta group law for gfx950 (y^2 = x^3 + 5 over Fp / Fq).
//
// Replaces pasta_curves 0.4.1 `pallas::{Affine,Point}` / `vesta::{Affine,Point}` (curves.rs)
// on the paths halo2_proofs' best_multiexp and the IPA commitment use (reference call sites
// /root/reference/src/test_utils.rs:12, 21, 40-49).
//
// Device accumulators use extended-Jacobian XYZZ coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2): a mixed add is 8M + 2S with no field inversion and the accumulator never
// has to be renormalised; identity is ZZ = 0.  Results cross the C ABI as Jacobian
// (X, Y, Z) -- the layout of pasta's `Point` -- normalised to Z = 1 (identity: all zero), so
// the bytes are unique and comparable limb-for-limb with the CPU path after `to_affine()`.
//
// Every formula carries the complete case analysis (identity operands, P + P, P + (-P)): the
// exceptional branches are taken for duplicate bases and in bucket running sums.
#pragma once
#include "field.h"

namespace trh {

// Memory-format PODs (what crosses the C ABI and what kernels keep in HBM / LDS): each field
// element is eight 32-bit words = pasta's [u64; 4].  AffineMem is the 64-byte base POD (the
// all-zero pattern, not on the curve, is the identity); JacobianMem is pasta's `Point` layout.
struct alignas(16) FeMem { u32 w[8]; };
struct alignas(16) AffineMem { FeMem x, y; };
struct alignas(16) XYZZMem { FeMem x, y, zz, zzz; };
struct alignas(16) JacobianMem { FeMem x, y, z; };

template <class F> TRH_HD Fe<F> fe_load(const FeMem& m) { return fe_load<F>(m.w); }
template <class F> TRH_HD void fe_store(const Fe<F>& a, FeMem& m) { fe_store(a, m.w); }

// register-form points
template <class F>
struct Affine {
    Fe<F> x, y;
};
template <class F>
struct XYZZ {
    Fe<F> x, y, zz, zzz;
};
template <class F>
struct Jacobian {
    Fe<F> x, y, z;
};

template <class F> TRH_HD bool aff_is_identity(const Affine<F>& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
template <class F> TRH_HD XYZZ<F> xyzz_identity() {
    XYZZ<F> r; r.x = fe_zero<F>(); r.y = fe_zero<F>(); r.zz = fe_zero<F>(); r.zzz = fe_zero<F>(); return r;
}
template <class F> TRH_HD bool xyzz_is_identity(const XYZZ<F>& p) { return fe_is_zero(p.zz); }
template <class F> TRH_HD XYZZ<F> xyzz_from_affine(const Affine<F>& p) {
    XYZZ<F> r;
    if (aff_is_identity(p)) return xyzz_identity<F>();
    r.x = p.x; r.y = p.y; r.zz = fe_one<F>(); r.zzz = fe_one<F>();
    return r;
}

// 2 * affine point (mdbl-2008-s-1), p != identity
template <class F> TRH_HD XYZZ<F> xyzz_dbl_affine(const Affine<F>& p) {
    XYZZ<F> r;
    Fe<F> U = fe_dbl(p.y);
    if (fe_is_zero(U)) return xyzz_identity<F>();  // order-2 point (none on these curves)
    Fe<F> V = fe_sqr(U), W = fe_mul(U, V), S = fe_mul(p.x, V);
    Fe<F> xx = fe_sqr(p.x), M = fe_add(fe_dbl(xx), xx);
    r.x = fe_sub(fe_sqr(M), fe_dbl(S));
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, p.y));
    r.zz = V; r.zzz = W;
    return r;
}

// 2 * P (dbl-2008-s-1, a = 0)
template <class F> TRH_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
    if (xyzz_is_identity(p)) return p;
    XYZZ<F> r;
    Fe<F> U = fe_dbl(p.y);
    Fe<F> V = fe_sqr(U), W = fe_mul(U, V), S = fe_mul(p.x, V);
    Fe<F> xx = fe_sqr(p.x), M = fe_add(fe_dbl(xx), xx);
    r.x = fe_sub(fe_sqr(M), fe_dbl(S));
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, p.y));
    r.zz = fe_mul(V, p.zz);
    r.zzz = fe_mul(W, p.zzz);
    return r;
}

// acc += p (madd-2008-s: 8M + 2S), p affine
template <class F> TRH_HD void xyzz_madd(XYZZ<F>& acc, const Affine<F>& p) {
    if (aff_is_identity(p)) return;
    if (xyzz_is_identity(acc)) { acc = xyzz_from_affine(p); return; }
    Fe<F> U2 = fe_mul(p.x, acc.zz), S2 = fe_mul(p.y, acc.zzz);
    Fe<F> P = fe_sub(U2, acc.x), R = fe_sub(S2, acc.y);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) acc = xyzz_dbl_affine(p);  // acc == p
        else acc = xyzz_identity<F>();                // acc == -p
        return;
    }
    Fe<F> PP = fe_sqr(P), PPP = fe_mul(P, PP), Q = fe_mul(acc.x, PP);
    Fe<F> x3 = fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q));
    acc.y = fe_sub(fe_mul(R, fe_sub(Q, x3)), fe_mul(acc.y, PPP));
    acc.x = x3;
    acc.zz = fe_mul(acc.zz, PP);
    acc.zzz = fe_mul(acc.zzz, PPP);
}

// a + b (add-2008-s: 12M + 2S)
template <class F> TRH_HD XYZZ<F> xyzz_add(const XYZZ<F>& a, const XYZZ<F>& b) {
    if (xyzz_is_identity(a)) return b;
    if (xyzz_is_identity(b)) return a;
    Fe<F> U1 = fe_mul(a.x, b.zz), U2 = fe_mul(b.x, a.zz);
    Fe<F> S1 = fe_mul(a.y, b.zzz), S2 = fe_mul(b.y, a.zzz);
    Fe<F> P = fe_sub(U2, U1), R = fe_sub(S2, S1);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return xyzz_dbl(a);
        return xyzz_identity<F>();
    }
    Fe<F> PP = fe_sqr(P), PPP = fe_mul(P, PP), Q = fe_mul(U1, PP);
    XYZZ<F> r;
    r.x = fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q));
    r.y = fe_sub(fe_mul(R, fe_sub(Q, r.x)), fe_mul(S1, PPP));
    r.zz = fe_mul(fe_mul(a.zz, b.zz), PP);
    r.zzz = fe_mul(fe_mul(a.zzz, b.zzz), PPP);
    return r;
}

template <class F> TRH_HD Affine<F> aff_neg(const Affine<F>& p) {
    Affine<F> r; r.x = p.x; r.y = fe_neg(p.y);  // identity (0,0) stays (0,0)
    return r;
}
template <class F> TRH_HD XYZZ<F> xyzz_neg(const XYZZ<F>& p) {
    XYZZ<F> r = p; r.y = fe_neg(p.y);
    return r;
}

// k * p for a small non-negative integer k (double-and-add, variable time)
template <class F> TRH_HD XYZZ<F> xyzz_mul_small(const XYZZ<F>& p, u32 k) {
    XYZZ<F> acc = xyzz_identity<F>();
    for (int i = 31; i >= 0; --i) {
        acc = xyzz_dbl(acc);
        if ((k >> i) & 1u) acc = xyzz_add(acc, p);
    }
    return acc;
}

// XYZZ -> affine (one inversion)
template <class F> TRH_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
    Affine<F> r;
    if (xyzz_is_identity(p)) { r.x = fe_zero<F>(); r.y = fe_zero<F>(); return r; }
    // 1/ZZZ, then 1/ZZ = ZZZ^-2 * ZZ^2  (ZZ^3 = ZZZ^2)
    Fe<F> zzz_inv = fe_inv(p.zzz);
    Fe<F> zz_inv = fe_mul(fe_sqr(zzz_inv), fe_sqr(p.zz));
    r.x = fe_mul(p.x, zz_inv);
    r.y = fe_mul(p.y, zzz_inv);
    return r;
}
// Jacobian with Z = 1 (identity: all zero) -- the normalised ABI output
template <class F> TRH_HD Jacobian<F> jac_from_affine(const Affine<F>& a) {
    Jacobian<F> j;
    j.x = a.x; j.y = a.y;
    j.z = aff_is_identity(a) ? fe_zero<F>() : fe_one<F>();
    return j;
}
// Jacobian (any Z) -> XYZZ: ZZ = Z^2, ZZZ = Z^3
template <class F> TRH_HD XYZZ<F> xyzz_from_jacobian(const Jacobian<F>& j) {
    XYZZ<F> r;
    if (fe_is_zero(j.z)) return xyzz_identity<F>();
    r.x = j.x; r.y = j.y; r.zz = fe_sqr(j.z); r.zzz = fe_mul(r.zz, j.z);
    return r;
}

// ---------------------------------------------------------------------------------------
// Lazy-domain points (field.h "Signed lazy domain with 29-bit limbs"): the MSM's bucket arithmetic.
// Coordinates are NORMALISED Fy values (signed, |x| < 4 m, |y| < 1.5 m, zz, zzz in (-0.02 m, 1.02 m) -- loose enough: the
// arithmetic tolerates 16 m); identity <=> zz is exactly zero.  AffineZ coordinates are in [0, 1.01 m); identity <=> x and y
// exactly zero.  Differences that only feed one multiplication stay lazy (no carry chain), and every y3 = A B - C D shares one
// Montgomery reduction (fy_mul2).  The exceptional cases of the full addition (P + P, P + (-P)) are a doubling / the identity in this domain too.
// ---------------------------------------------------------------------------------------
template <class F>
struct AffineZ {
    Fy<F> x, y;
};
template <class F>
struct XYZZz {
    Fy<F> x, y, zz, zzz;
};
struct alignas(16) XYZZzMem { u32 w[36]; };  // raw limbs of x, y, zz, zzz (accumulate -> combine scratch)

template <class F> TRH_HD XYZZz<F> xyzzz_identity() {
    XYZZz<F> r; r.x = fy_zero<F>(); r.y = fy_zero<F>(); r.zz = fy_zero<F>(); r.zzz = fy_zero<F>(); return r;
}
template <class F> TRH_HD bool xyzzz_is_identity(const XYZZz<F>& p) { return fy_is_exact_zero(p.zz); }
template <class F> TRH_HD XYZZ<F> xyzzz_to_canonical(const XYZZz<F>& p) {
    XYZZ<F> r;
    if (xyzzz_is_identity(p)) return xyzz_identity<F>();
    r.x = fy_to_fe(p.x); r.y = fy_to_fe(p.y); r.zz = fy_to_fe(p.zz); r.zzz = fy_to_fe(p.zzz);
    return r;
}
template <class F> TRH_HD XYZZz<F> xyzzz_from_canonical(const XYZZ<F>& p) {
    XYZZz<F> r;
    if (xyzz_is_identity(p)) return xyzzz_identity<F>();
    r.x = fy_from_fe(p.x); r.y = fy_from_fe(p.y); r.zz = fy_from_fe(p.zz); r.zzz = fy_from_fe(p.zzz);
    return r;
}

// 2 p for an affine point of the lazy domain (not the identity; y != 0: no points of order two): dbl-2008-s-1 with Z = 1
template <class F> TRH_HD XYZZz<F> xyzzz_dbl_affine(const AffineZ<F>& p) {
    XYZZz<F> r;
    const Fy<F> U = fy_add(p.y, p.y);
    const Fy<F> V = fy_sqr(U), W = fy_mul(U, V), S = fy_mul(p.x, V);
    const Fy<F> xx = fy_sqr(p.x), M = fy_add(fy_add(xx, xx), xx);
    r.x = fy_sub_sub2(fy_sqr(M), fy_zero<F>(), S);                                        // M^2 - 2 S
    r.y = fy_mul2(fy_sub_lazy(S, r.x), M, fy_neg_lazy(W), p.y);                           // M (S - x3) - W y
    r.zz = V; r.zzz = W;
    return r;
}

// 2 p for a lazy XYZZ point (dbl-2008-s-1, a = 0)
template <class F> TRH_HD XYZZz<F> xyzzz_dbl(const XYZZz<F>& p) {
    if (xyzzz_is_identity(p)) return p;
    XYZZz<F> r;
    const Fy<F> U = fy_add(p.y, p.y);
    const Fy<F> V = fy_sqr(U), W = fy_mul(U, V), S = fy_mul(p.x, V);
    const Fy<F> xx = fy_sqr(p.x), M = fy_add(fy_add(xx, xx), xx);
    r.x = fy_sub_sub2(fy_sqr(M), fy_zero<F>(), S);
    r.y = fy_mul2(fy_sub_lazy(S, r.x), M, fy_neg_lazy(W), p.y);
    r.zz = fy_mul(V, p.zz);
    r.zzz = fy_mul(W, p.zzz);
    return r;
}

// acc += p, p affine in the lazy domain: 9 reductions for the 8M + 2S of madd-2008-s
template <class F> TRH_HD void xyzzz_madd(XYZZz<F>& acc, const AffineZ<F>& p) {
    if (fy_is_exact_zero(p.x) && fy_is_exact_zero(p.y)) return;
    if (xyzzz_is_identity(acc)) { acc.x = p.x; acc.y = p.y; acc.zz = fy_one<F>(); acc.zzz = fy_one<F>(); return; }
    const Fy<F> P = fy_mul_sub(p.x, acc.zz, acc.x), R = fy_mul_sub(p.y, acc.zzz, acc.y);  // U2 - X, S2 - Y, normalised: both are squared
    if (fy_is_zero_mod(P)) {  // same x: acc is p (the sum is 2 p, which only needs p) or -p (the sum is the identity)
        if (fy_is_zero_mod(R)) acc = xyzzz_dbl_affine(p);
        else acc = xyzzz_identity<F>();
        return;
    }
    const Fy<F> PP = fy_sqr(P), PPP = fy_mul(P, PP), Q = fy_mul(acc.x, PP);
    const Fy<F> x3 = fy_sqr_sub_sub2(R, PPP, Q);                                           // R^2 - PPP - 2 Q
    acc.y = fy_mul2(fy_sub_lazy(Q, x3), R, fy_neg_lazy(acc.y), PPP);                       // R (Q - x3) - Y PPP, one reduction
    acc.x = x3;
    acc.zz = fy_mul(acc.zz, PP);
    acc.zzz = fy_mul(acc.zzz, PPP);
}

// The generic case of acc += p -- neither operand is the identity -- as ONE straight line that updates acc in place.  Returns P = 0
// (mod m), i.e. "same x": acc was p or -p and now holds garbage; the caller patches those lanes (R = 0 (mod m) tells p + p from
// p - p).  Keeping the special cases out of this function keeps their register copies out of the accumulation loop: with the
// branches inside (xyzzz_madd above) every merge point copied the 36 accumulator limbs, ~220 moves per mixed addition.
template <class F> TRH_HD bool xyzzz_madd_main(XYZZz<F>& acc, const AffineZ<F>& p, Fy<F>& R) {
    const Fy<F> P = fy_mul_sub(p.x, acc.zz, acc.x);
    R = fy_mul_sub(p.y, acc.zzz, acc.y);
    bool same_x = false;
    if (fy_maybe_zero_mod(P)) {  // a real branch: the full comparison (~50 instructions) must not be flattened into the line
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");
#endif
        same_x = fy_is_zero_mod(P);
    }
    const Fy<F> PP = fy_sqr(P), PPP = fy_mul(P, PP), Q = fy_mul(acc.x, PP);
    const Fy<F> x3 = fy_sqr_sub_sub2(R, PPP, Q);
    acc.y = fy_mul2(fy_sub_lazy(Q, x3), R, fy_neg_lazy(acc.y), PPP);
    acc.x = x3;
    acc.zz = fy_mul(acc.zz, PP);
    acc.zzz = fy_mul(acc.zzz, PPP);
    return same_x;
}

// a + b, both lazy XYZZ
template <class F> TRH_HD XYZZz<F> xyzzz_add(const XYZZz<F>& a, const XYZZz<F>& b) {
    if (xyzzz_is_identity(a)) return b;
    if (xyzzz_is_identity(b)) return a;
    const Fy<F> U1 = fy_mul(a.x, b.zz), U2 = fy_mul(b.x, a.zz);
    const Fy<F> S1 = fy_mul(a.y, b.zzz), S2 = fy_mul(b.y, a.zzz);
    const Fy<F> P = fy_sub(U2, U1), R = fy_sub(S2, S1);
    if (fy_is_zero_mod(P)) {
        // same x: b is a (the sum is 2 a) or -a (the identity).  In this domain (round 6): through the canonical formulas every inlined copy of
        // this function carried ~11 000 instructions of conversions and canonical arithmetic for a case that almost never runs -- msm_reduce_kernel
        // was 84 581 instructions long, 660 KiB against a 64 KiB instruction cache
        if (fy_is_zero_mod(R)) return xyzzz_dbl(a);
        return xyzzz_identity<F>();
    }
    const Fy<F> PP = fy_sqr(P), PPP = fy_mul(P, PP), Q = fy_mul(U1, PP);
    XYZZz<F> r;
    r.x = fy_sqr_sub_sub2(R, PPP, Q);
    r.y = fy_mul2(fy_sub_lazy(Q, r.x), R, fy_neg_lazy(S1), PPP);
    r.zz = fy_mul(fy_mul(a.zz, b.zz), PP);
    r.zzz = fy_mul(fy_mul(a.zzz, b.zzz), PPP);
    return r;
}

template <class F> TRH_HD Affine<F> aff_load(const AffineMem& m) {
    Affine<F> r; r.x = fe_load<F>(m.x); r.y = fe_load<F>(m.y); return r;
}
template <class F> TRH_HD void aff_store(const Affine<F>& a, AffineMem& m) { fe_store(a.x, m.x); fe_store(a.y, m.y); }
template <class F> TRH_HD XYZZ<F> xyzz_load(const XYZZMem& m) {
    XYZZ<F> r; r.x = fe_load<F>(m.x); r.y = fe_load<F>(m.y); r.zz = fe_load<F>(m.zz); r.zzz = fe_load<F>(m.zzz); return r;
}
template <class F> TRH_HD void xyzz_store(const XYZZ<F>& a, XYZZMem& m) {
    fe_store(a.x, m.x); fe_store(a.y, m.y); fe_store(a.zz, m.zz); fe_store(a.zzz, m.zzz);
}
template <class F> TRH_HD Jacobian<F> jac_load(const JacobianMem& m) {
    Jacobian<F> r; r.x = fe_load<F>(m.x); r.y = fe_load<F>(m.y); r.z = fe_load<F>(m.z); return r;
}
template <class F> TRH_HD void jac_store(const Jacobian<F>& a, JacobianMem& m) { fe_store(a.x, m.x); fe_store(a.y, m.y); fe_store(a.z, m.z); }

}  // namespace trh
