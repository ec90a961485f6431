// CPU unit test of the signed 29-bit lazy domain (csrc/field.h "Fy", csrc/curve.h "XYZZz": the MSM's bucket arithmetic) against the
// canonical nine-limb Montgomery implementation the rest of the library uses: conversions, products, merged products, and long
// chains of mixed additions / additions / doublings with negated bases, identities, P + P and P + (-P), with the magnitude
// invariants checked after every step.  Built with -fsanitize=undefined too (tests/test_hostcombine.py): a signed overflow of a
// 32-bit limb or a 64-bit column would be reported there.   g++ only, no GPU.
#include <cstdio>
#include <cstring>
#include "../../tiny-ram-halo2_amd/csrc/curve.h"
using namespace trh;

static u64 seed = 0x9e3779b97f4a7c15ull;
static u64 next() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; }

template <class F> static Fe<F> rand_fe() {
    u32 w[8];
    for (int i = 0; i < 8; ++i) w[i] = (u32)next();
    w[7] &= 0x3fffffffu;  // < 2^254 < m
    return fe_load<F>(w);
}
template <class F> static bool same(const Fe<F>& a, const Fe<F>& b) { return fe_eq(a, b); }
template <class F> static bool normalised(const Fy<F>& a, int top_bits) {
    for (int i = 0; i < 8; ++i) if (a.l[i] < 0 || a.l[i] > YMASK) return false;
    const i32 t = a.l[8] < 0 ? -a.l[8] : a.l[8];
    return t < (1 << top_bits);
}
template <class F> static bool same_point(const XYZZz<F>& z, const XYZZ<F>& c) {
    const Affine<F> a = xyzz_to_affine(xyzzz_to_canonical(z)), b = xyzz_to_affine(c);
    return same(a.x, b.x) && same(a.y, b.y);
}
template <class F> static bool small_limbs(const Fy<F>& a, int top_bits) {  // what a negated base coordinate looks like: |l[k]| < 2^29
    for (int i = 0; i < 8; ++i) if (a.l[i] < -YMASK || a.l[i] > YMASK) return false;
    const i32 t = a.l[8] < 0 ? -a.l[8] : a.l[8];
    return t < (1 << top_bits);
}
template <class F> static bool in_bounds(const XYZZz<F>& p) {  // |x| < 8 m, |y| < 4 m, zz / zzz within (-m/4, 5 m / 4): top limb of k m is k 2^22
    return normalised(p.x, 25) && small_limbs(p.y, 24) && normalised(p.zz, 23) && normalised(p.zzz, 23);
}

template <class F>
static int run(const char* name) {
    int bad = 0;
    auto fail = [&](const char* what, int i) { ++bad; if (bad < 10) std::printf("%s: %s (case %d)\n", name, what, i); };
    // ---- field layer ----
    for (int i = 0; i < 2000; ++i) {
        const Fe<F> a = rand_fe<F>(), b = rand_fe<F>(), c = rand_fe<F>(), d = rand_fe<F>();
        const Fy<F> ya = fy_from_fe(a), yb = fy_from_fe(b), yc = fy_from_fe(c), yd = fy_from_fe(d);
        if (!normalised(ya, 23) || ya.l[8] < 0) fail("from_fe not normalised / negative", i);
        if (!same(fy_to_fe(ya), a)) fail("to_fe(from_fe(a)) != a", i);
        if (!same(fy_to_fe(fy_mul(ya, yb)), fe_mul(a, b))) fail("mul", i);
        if (!same(fy_to_fe(fy_sqr(ya)), fe_sqr(a))) fail("sqr", i);
        if (!same(fy_to_fe(fy_mul2(ya, yb, yc, yd)), fe_add(fe_mul(a, b), fe_mul(c, d)))) fail("mul2", i);
        if (!same(fy_to_fe(fy_mul2(fy_sub_lazy(ya, yc), yb, fy_neg_lazy(yc), yd)), fe_sub(fe_mul(fe_sub(a, c), b), fe_mul(c, d)))) fail("mul2 lazy / negated", i);
        if (!same(fy_to_fe(fy_sub(ya, yb)), fe_sub(a, b)) || !same(fy_to_fe(fy_add(ya, yb)), fe_add(a, b))) fail("add / sub", i);
        if (!same(fy_to_fe(fy_sub_sub2(ya, yb, yc)), fe_sub(fe_sub(a, b), fe_dbl(c)))) fail("sub_sub2", i);
        if (!same(fy_to_fe(fy_mul_sub(ya, yb, yc)), fe_sub(fe_mul(a, b), c))) fail("mul_sub", i);
        if (!same(fy_to_fe(fy_sqr_sub_sub2(ya, yb, yc)), fe_sub(fe_sub(fe_sqr(a), b), fe_dbl(c))) || !normalised(fy_sqr_sub_sub2(ya, yb, yc), 25)) fail("sqr_sub_sub2", i);
        if (!same(fy_to_fe(fy_mul(fy_sub_lazy(ya, yb), yc)), fe_mul(fe_sub(a, b), c))) fail("lazy operand", i);
        // zero test: k m for small |k| and near misses
        Fy<F> mm;
        for (int k = 0; k < NLIMBS; ++k) mm.l[k] = ymod_limb<F>(k);
        Fy<F> acc = fy_zero<F>();
        const int kk = (int)(next() % 9);
        for (int k = 0; k < kk; ++k) acc = fy_add(acc, mm);
        if (!fy_is_zero_mod(acc) || !fy_is_zero_mod(fy_sub(fy_zero<F>(), acc))) fail("k m not recognised as zero", i);
        Fy<F> off = acc; off.l[3] ^= 1;
        if (fy_is_zero_mod(off) || fy_is_zero_mod(ya) != fe_is_zero(a)) fail("zero test false positive", i);
        // memory round trip
        u32 w[8];
        fy_store(ya, w);
        const Fy<F> back = fy_load<F>(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
        if (memcmp(&back, &ya, sizeof(back)) != 0) fail("store / load", i);
    }
    // ---- curve layer: a long accumulation with every exceptional case, both implementations side by side ----
    Affine<F> G; G.x = fe_neg(fe_one<F>()); G.y = fe_dbl(fe_one<F>());
    const int NB = 64;
    Affine<F> bases[NB];
    AffineZ<F> bz[NB];
    {
        XYZZ<F> p = xyzz_from_affine(G);
        for (int i = 0; i < NB; ++i) {
            const int steps = 1 + (int)(next() % 3);
            for (int s2 = 0; s2 < steps; ++s2) { p = xyzz_dbl(p); if (next() & 1) xyzz_madd(p, G); }
            bases[i] = xyzz_to_affine(p);
            bz[i].x = fy_from_fe(bases[i].x); bz[i].y = fy_from_fe(bases[i].y);
        }
        bases[7].x = fe_zero<F>(); bases[7].y = fe_zero<F>(); bz[7].x = fy_zero<F>(); bz[7].y = fy_zero<F>();  // an identity base
    }
    XYZZ<F> c = xyzz_identity<F>();
    XYZZz<F> z = xyzzz_identity<F>();
    XYZZz<F> snapshot = z; XYZZ<F> csnap = c;
    for (int it = 0; it < 6000; ++it) {
        const u64 r = next();
        const int i = (int)(r % NB);
        const bool neg = (r >> 20) & 1;
        const u64 kind = (r >> 24) % 64;
        if (kind == 0) { z = xyzzz_identity<F>(); c = xyzz_identity<F>(); }          // restart (bucket boundary)
        else if (kind == 1) { z = xyzzz_dbl(z); c = xyzz_dbl(c); }
        else if (kind == 2) { z = xyzzz_add(z, snapshot); c = xyzz_add(c, csnap); }    // full addition with an earlier value
        else if (kind == 3) { z = xyzzz_add(z, z); c = xyzz_add(c, c); }               // P + P through the addition
        else if (kind == 4) { XYZZz<F> m = z; m.y = fy_sub(fy_zero<F>(), z.y); z = xyzzz_add(z, m); c = xyzz_identity<F>(); }  // P + (-P)
        else if (kind == 5) { snapshot = z; csnap = c; }
        else {
            Affine<F> b = bases[i];
            AffineZ<F> y = bz[i];
            if (neg) {
                b = aff_neg(b);
                for (int k = 0; k < NLIMBS; ++k) y.y.l[k] = -y.y.l[k];  // the kernel's limb-wise negation
            }
            if (kind == 6 && !xyzz_is_identity(c)) {  // make the accumulator equal to +-the base: the doubling / cancellation branches of madd
                z = xyzzz_identity<F>(); c = xyzz_identity<F>();
                xyzzz_madd(z, bz[i]); xyzz_madd(c, bases[i]);
            }
            xyzzz_madd(z, y);
            xyzz_madd(c, b);
        }
        if (!in_bounds(z)) { fail("magnitude invariant", it); break; }
        if ((it % 16) == 0 || kind < 8) if (!same_point(z, c) || xyzzz_is_identity(z) != xyzz_is_identity(c)) { fail("point mismatch", it); break; }
    }
    if (!same_point(z, c)) fail("final point", 0);

    // ---- the accumulation kernel's step (msm.hip: msm_accumulate_seg_kernel) restated on the host: the first point of a bucket is taken
    // over, the generic case runs as one straight line (xyzzz_madd_main), the same-x cases are patched afterwards, identity bases skip
    {
        XYZZ<F> cc = xyzz_identity<F>();
        XYZZz<F> acc = xyzzz_identity<F>();
        bool fresh = true;
        int last = -1;
        for (int it = 0; it < 6000; ++it) {
            const u64 r = next();
            int i = (int)(r % NB);
            bool neg = (r >> 20) & 1;
            const u64 kind = (r >> 24) % 32;
            if (kind == 0) { acc = xyzzz_identity<F>(); cc = xyzz_identity<F>(); fresh = true; last = -1; continue; }  // bucket boundary (the kernel leaves acc stale; fresh guards it)
            if (kind == 1 && last >= 0 && !fresh) { i = last; }                  // may repeat the previous base: same x when the bucket holds just that point
            if (kind == 2 && last >= 0) { acc = xyzzz_identity<F>(); cc = xyzz_identity<F>(); fresh = true; i = last; }  // restart, then base, then (often) +-base again
            Affine<F> b = bases[i];
            AffineZ<F> y = bz[i];
            if (neg) { b = aff_neg(b); for (int k = 0; k < NLIMBS; ++k) y.y.l[k] = -y.y.l[k]; }
            i32 yor = 0;
            for (int k = 0; k < NLIMBS; ++k) yor |= y.y.l[k];
            const bool p_identity = yor == 0;
            if (!p_identity) {
                if (fresh) { acc.x = y.x; acc.y = y.y; acc.zz = fy_one<F>(); acc.zzz = fy_one<F>(); fresh = false; }
                else {
                    Fy<F> R;
                    if (xyzzz_madd_main(acc, y, R)) {
                        if (fy_is_zero_mod(R)) acc = xyzzz_dbl_affine(y);
                        else { acc = xyzzz_identity<F>(); fresh = true; }
                    }
                }
            } else if (fresh) acc = xyzzz_identity<F>();
            xyzz_madd(cc, b);
            last = i;
            if (!fresh && !in_bounds(acc)) { fail("kernel step: magnitude invariant", it); break; }
            if (fresh != xyzz_is_identity(cc)) { fail("kernel step: fresh flag vs identity", it); break; }
            if (!fresh && ((it % 8) == 0 || kind < 3) && !same_point(acc, cc)) { fail("kernel step: point mismatch", it); break; }
        }
    }
    // ---- balanced constants and unnormalised multiplicands (the NTT's butterflies), non-negative products (stored tables) ----
    for (int i = 0; i < 2000; ++i) {
        const Fe<F> a = rand_fe<F>(), b = rand_fe<F>(), c2 = rand_fe<F>(), d = rand_fe<F>();
        const Fy<F> ya = fy_from_fe(a), yb = fy_from_fe(b), yc = fy_from_fe(c2), yd = fy_from_fe(d);
        const Fy<F> bal = fy_balance(yb);
        for (int k = 0; k < 8; ++k) if (bal.l[k] < -(1 << 28) || bal.l[k] >= (1 << 28)) fail("balance range", i);
        if (!same(fy_to_fe(bal), b)) fail("balance value", i);
        const Fy<F> lvl3 = fy_sub_lazy(fy_add_lazy(ya, yc), yd);   // |limb| < 3 * 2^29, never normalised
        if (!same(fy_to_fe(fy_mul(lvl3, bal)), fe_mul(fe_sub(fe_add(a, c2), d), b))) fail("level-3 multiplicand x balanced constant", i);
        const Fy<F> nn = fy_mul_nonneg(ya, yb);
        if (!normalised(nn, 24) || nn.l[8] < 0 || !same(fy_to_fe(nn), fe_mul(a, b))) fail("mul_nonneg", i);
    }
    return bad;
}

int main() {
    const int bad = run<FpParams>("fp") + run<FqParams>("fq");
    std::printf("lazy29: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
