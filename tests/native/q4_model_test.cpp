// CPU model of the quad-lane group law of csrc/curve_q4.h: the same sequence of field operations, lane by lane, with the DPP exchanges written
// as array indexing -- against curve.h's xyzzz_add / xyzzz_dbl on random points, identities, P + P and P + (-P).  What this pins is the
// ALGORITHM (which lane multiplies what, in which step, and the magnitude bounds: built with -fsanitize=undefined, a signed overflow of a limb
// or of a 64-bit column is reported); the device code's exchanges are tested on the GPU (tests/test_gpu_q4.py).   g++ only, no GPU.
#include <cstdio>
#include <cstring>
#include "../../tiny-ram-halo2_amd/csrc/curve.h"
using namespace trh;

static u64 seed = 0x243f6a8885a308d3ull;
static u64 next() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; }

template <class F> struct Quad { Fy<F> l[4]; };
template <class F> static Quad<F> perm(const Quad<F>& v, int p0, int p1, int p2, int p3) { Quad<F> r; r.l[0] = v.l[p0]; r.l[1] = v.l[p1]; r.l[2] = v.l[p2]; r.l[3] = v.l[p3]; return r; }
template <class F> static Quad<F> spread(const XYZZz<F>& p) { Quad<F> r; r.l[0] = p.x; r.l[1] = p.y; r.l[2] = p.zz; r.l[3] = p.zzz; return r; }
template <class F> static XYZZz<F> gather(const Quad<F>& q) { XYZZz<F> p; p.x = q.l[0]; p.y = q.l[1]; p.zz = q.l[2]; p.zzz = q.l[3]; return p; }

template <class F> static Quad<F> q4_dbl_model(const Quad<F>& A) {
    const bool id = fy_is_exact_zero(A.l[2]);
    Quad<F> U, E1, F2, M, G3, X3, H, r;
    for (int q = 0; q < 4; ++q) U.l[q] = fy_add(A.l[1], A.l[1]);
    for (int q = 0; q < 4; ++q) E1.l[q] = fy_sqr(q == 0 ? A.l[q] : U.l[q]);
    const Quad<F> E1p = perm(E1, 1, 1, 2, 3);
    for (int q = 0; q < 4; ++q) F2.l[q] = fy_mul((q & 1) ? U.l[q] : A.l[q], E1p.l[q]);
    for (int q = 0; q < 4; ++q) M.l[q] = fy_add(fy_add(E1.l[q], E1.l[q]), E1.l[q]);
    for (int q = 0; q < 4; ++q) G3.l[q] = fy_mul(q == 0 ? M.l[q] : F2.l[q], q == 0 ? M.l[q] : A.l[q]);
    for (int q = 0; q < 4; ++q) X3.l[q] = fy_sub_sub2(G3.l[q], fy_zero<F>(), F2.l[q]);
    for (int q = 0; q < 4; ++q) H.l[q] = fy_mul(fy_sub_lazy(F2.l[q], X3.l[q]), M.l[q]);
    const Quad<F> H0 = perm(H, 0, 0, 0, 0);
    for (int q = 0; q < 4; ++q) { const Fy<F> Y3 = fy_sub(H0.l[q], G3.l[q]); r.l[q] = q == 0 ? X3.l[q] : q == 1 ? Y3 : q == 2 ? F2.l[q] : G3.l[q]; }
    return id ? A : r;
}
template <class F> static Quad<F> q4_add_model(const Quad<F>& A, const Quad<F>& B) {
    const bool idA = fy_is_exact_zero(A.l[2]), idB = fy_is_exact_zero(B.l[2]);
    Quad<F> T1, T2, D, E, F4, X3, G, r;
    const Quad<F> Bs = perm(B, 2, 3, 2, 3), As = perm(A, 2, 3, 2, 3);
    for (int q = 0; q < 4; ++q) { T1.l[q] = fy_mul(A.l[q], Bs.l[q]); T2.l[q] = fy_mul(B.l[q], As.l[q]); D.l[q] = fy_sub(T2.l[q], T1.l[q]); }
    const bool same_x = fy_is_zero_mod(D.l[0]) && !idA && !idB, same_y = fy_is_zero_mod(D.l[1]);
    const Quad<F> Dq = perm(D, 0, 1, 0, 0), D0 = perm(D, 0, 0, 0, 0);
    for (int q = 0; q < 4; ++q) E.l[q] = fy_sqr(Dq.l[q]);
    const Quad<F> Eb = perm(E, 0, 0, 2, 3);
    for (int q = 0; q < 4; ++q) F4.l[q] = fy_mul((q & 1) ? D0.l[q] : T1.l[q], Eb.l[q]);
    const Quad<F> E1 = perm(E, 1, 1, 1, 1), F1 = perm(F4, 1, 1, 1, 1), F0 = perm(F4, 0, 0, 0, 0);
    for (int q = 0; q < 4; ++q) X3.l[q] = fy_sub_sub2(E1.l[q], F1.l[q], F0.l[q]);
    const Quad<F> T1p = perm(T1, 1, 1, 2, 3), F4p = perm(F4, 1, 1, 2, 3);
    for (int q = 0; q < 4; ++q) G.l[q] = fy_mul(q == 1 ? D.l[q] : T1p.l[q], q == 1 ? fy_sub_lazy(F0.l[q], X3.l[q]) : F4p.l[q]);
    const Quad<F> G0 = perm(G, 0, 0, 0, 0);
    for (int q = 0; q < 4; ++q) { const Fy<F> Y3 = fy_sub(G.l[q], G0.l[q]); r.l[q] = q == 0 ? X3.l[q] : q == 1 ? Y3 : q == 2 ? F4.l[q] : G.l[q]; }
    if (same_x) { if (same_y) r = q4_dbl_model(A); else for (int q = 0; q < 4; ++q) r.l[q] = fy_zero<F>(); }
    return idA ? B : idB ? A : r;
}

template <class F> static Fe<F> rand_fe() { u32 w[8]; for (int i = 0; i < 8; ++i) w[i] = (u32)next(); w[7] &= 0x3fffffffu; return fe_load<F>(w); }
template <class F> static bool same_point(const XYZZz<F>& z, const XYZZz<F>& c) {
    const Affine<F> a = xyzz_to_affine(xyzzz_to_canonical(z)), b = xyzz_to_affine(xyzzz_to_canonical(c));
    return fe_eq(a.x, b.x) && fe_eq(a.y, b.y);
}
template <class F> static int run(const char* name) {
    int bad = 0;
    // a point of the curve: G = (-1, 2) and multiples by a running chain (canonical arithmetic)
    Affine<F> G; G.x = fe_neg(fe_one<F>()); G.y = fe_dbl(fe_one<F>());
    XYZZ<F> cur = xyzz_from_affine(G);
    XYZZz<F> acc = xyzzz_from_canonical(cur), other = xyzzz_dbl(acc);
    for (int i = 0; i < 400; ++i) {
        // general sums with non-trivial ZZ / ZZZ on both sides
        const XYZZz<F> want = xyzzz_add(acc, other);
        const XYZZz<F> got = gather(q4_add_model(spread(acc), spread(other)));
        if (!same_point(got, want)) { if (++bad < 8) std::printf("%s: add (case %d)\n", name, i); }
        const XYZZz<F> wd = xyzzz_dbl(acc), gd = gather(q4_dbl_model(spread(acc)));
        if (!same_point(gd, wd)) { if (++bad < 8) std::printf("%s: dbl (case %d)\n", name, i); }
        // P + P, P + (-P), identities
        XYZZz<F> neg = acc; neg.y = fy_norm(fy_neg_lazy(acc.y));
        if (!same_point(gather(q4_add_model(spread(acc), spread(acc))), wd)) { if (++bad < 8) std::printf("%s: P + P (case %d)\n", name, i); }
        if (!xyzzz_is_identity(gather(q4_add_model(spread(acc), spread(neg))))) { if (++bad < 8) std::printf("%s: P - P (case %d)\n", name, i); }
        const XYZZz<F> idp = xyzzz_identity<F>();
        if (!same_point(gather(q4_add_model(spread(idp), spread(acc))), acc) || !same_point(gather(q4_add_model(spread(acc), spread(idp))), acc)) { if (++bad < 8) std::printf("%s: identity operand (case %d)\n", name, i); }
        if (!xyzzz_is_identity(gather(q4_add_model(spread(idp), spread(idp)))) || !xyzzz_is_identity(gather(q4_dbl_model(spread(idp))))) { if (++bad < 8) std::printf("%s: identity + identity (case %d)\n", name, i); }
        // walk on: the quad results feed the next round, so that representations produced by the quad form are also its inputs
        other = gather(q4_add_model(spread(other), spread(got)));
        acc = (i & 1) ? gd : got;
    }
    return bad;
}
int main() {
    const int bad = run<FpParams>("Fp") + run<FqParams>("Fq");
    std::printf(bad ? "q4 model: FAILED (%d)\n" : "q4 model: ok\n", bad);
    return bad ? 1 : 0;
}
