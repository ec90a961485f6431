// Quad-lane group law for gfx950: ONE XYZZ point spread over the four lanes of a DPP quad -- lane q = lane & 3 holds coordinate q of
// (x, y, zz, zzz) as a normalised Fy -- so that the independent field multiplications of a point operation run side by side instead of one
// after the other, with the operands exchanged by `v_mov_b32 ... quad_perm` (one instruction per limb, no LDS).
//
// Why: the tail of every small MSM -- a lone fixed-base commitment, an IPA round -- is ONE dependent chain of ~45 point operations in the
// bucket reduction (running sums over a slice, the slice offset by double-and-add, the workgroup tree), executed by waves that have a SIMD to
// themselves.  Such a chain runs at the instruction count of a point operation (add-2008-s: 12 M + 2 S, ~1900 instructions one after the
// other); nothing else on the chip shortens it.  In the quad form the same addition is FIVE multiplication steps deep (6 + 2 + 3 + 3 products
// over four lanes), the doubling (dbl-2008-s-1, a = 0) four: ~2 x shorter chains for 4 x the lanes, which a latency-bound launch has to spare.
// (north_star's "wavefront-level shuffle reduction", applied where it pays: inside the point operation.)
//
// Replaces nothing in the reference by itself: it is the arithmetic of pasta_curves' `Point + Point` / `double()` (curves.rs) as used by
// best_multiexp's bucket sums, with the complete case analysis (identity operands, P + P, P + (-P)); results are the same group elements as
// curve.h's xyzzz_add / xyzzz_dbl (tests: every MSM parity test with TRH_REDUCE_Q4=1 / 0, tests/test_gpu_q4.py).
#pragma once
#include "curve.h"

namespace trh {

#if defined(__HIPCC__)
// The exchange stays a `v_mov_b32 ... quad_perm` of its own: the empty asm statement keeps the compiler's DPP combiner from folding it into
// the instruction that uses the value.  Folded into a subtraction whose SECOND operand is the exchanged value (`g - perm(g)`), ROCm 7.2's
// combiner emitted the operands the other way round (tools/q4_dev_test.hip: Y3 = R (Q - X3) - S1 PPP came out negated while every
// other intermediate matched the CPU model) -- and a separate move is what the cost model of this file assumes anyway.
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ i32 q4_perm_i32(i32 v) {
    i32 r = __builtin_amdgcn_update_dpp(0, v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
#ifndef TRH_TEST_DROP_Q4_WORKAROUND  // (tests/native/libtrh_q4broken.so is built without the statement: trh_init's self-test must refuse that library)
    asm volatile("" : "+v"(r));
#endif
    return r;
}
// lane i of every quad reads the value of lane P_i of its quad
template <class F, int P0, int P1, int P2, int P3>
__device__ __forceinline__ Fy<F> q4_perm(const Fy<F>& v) {
    Fy<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = q4_perm_i32<P0, P1, P2, P3>(v.l[i]);
    return r;
}
template <class F>
__device__ __forceinline__ Fy<F> q4_select(bool c, const Fy<F>& a, const Fy<F>& b) {
    Fy<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
// the point is the identity <=> its zz (lane 2) is exactly zero
template <class F>
__device__ __forceinline__ bool q4_is_identity(const Fy<F>& A) { return q4_perm_i32<2, 2, 2, 2>(fy_is_exact_zero(A) ? 1 : 0) != 0; }

// 2 A (dbl-2008-s-1, a = 0): four multiplication steps
//   step 1   l0: xx = x^2                 l1..3: V = U^2, U = 2 y
//   step 2   l0: S = x V      l1: W = U V      l2: ZZ3 = zz V      l3: W
//   step 3   l0: M^2 (M = 3 xx)           l1: W y              l3: ZZZ3 = W zzz
//   step 4   l0: M (S - X3),  X3 = M^2 - 2 S;   then l1: Y3 = M (S - X3) - W y
template <class F>
__device__ __forceinline__ Fy<F> q4_dbl(const Fy<F>& A, const int q) {
    const bool id = q4_is_identity(A);
    const Fy<F> Y = q4_perm<F, 1, 1, 1, 1>(A);
    const Fy<F> U = fy_add(Y, Y);
    const Fy<F> E1 = fy_sqr(q4_select(q == 0, A, U));
    const Fy<F> F2 = fy_mul(q4_select((q & 1) != 0, U, A), q4_perm<F, 1, 1, 2, 3>(E1));
    const Fy<F> M = fy_add(fy_add(E1, E1), E1);
    const Fy<F> G3 = fy_mul(q4_select(q == 0, M, F2), q4_select(q == 0, M, A));
    const Fy<F> X3 = fy_sub_sub2(G3, fy_zero<F>(), F2);
    const Fy<F> H = fy_mul(fy_sub_lazy(F2, X3), M);
    const Fy<F> Y3 = fy_sub(q4_perm<F, 0, 0, 0, 0>(H), G3);
    const Fy<F> r = q4_select(q == 0, X3, q4_select(q == 1, Y3, q4_select(q == 2, F2, G3)));
    return q4_select(id, A, r);
}

// A + B (add-2008-s): five multiplication steps
//   step 1   T1 = A * B[2,3,2,3]:   l0: U1 = x1 zz2    l1: S1 = y1 zzz2    l2: zz1 zz2     l3: zzz1 zzz2
//   step 2   T2 = B * A[2,3,2,3]:   l0: U2 = x2 zz1    l1: S2 = y2 zzz1;   D = T2 - T1:    l0: P, l1: R
//   step 3   E = D[0,1,0,0]^2:      l0: PP             l1: RR              l2, l3: PP
//   step 4   l0: Q = U1 PP          l1: PPP = P PP     l2: ZZ3 = zz1 zz2 PP                l3: PPP;   X3 = RR - PPP - 2 Q (every lane)
//   step 5   l0: S1 PPP             l1: R (Q - X3)     l3: ZZZ3 = zzz1 zzz2 PPP;           l1: Y3 = R (Q - X3) - S1 PPP
// Special cases: an identity operand returns the other one; P = 0 (same x) is a doubling when R = 0 as well and the identity otherwise -- the
// doubling runs behind a wave-uniform vote, so the straight line above carries no copy of it.
template <class F>
__device__ __forceinline__ Fy<F> q4_add(const Fy<F>& A, const Fy<F>& B, const int q) {
    const bool idA = q4_is_identity(A), idB = q4_is_identity(B);
    const Fy<F> T1 = fy_mul(A, q4_perm<F, 2, 3, 2, 3>(B));
    const Fy<F> T2 = fy_mul(B, q4_perm<F, 2, 3, 2, 3>(A));
    const Fy<F> D = fy_sub(T2, T1);
    int zero_mod = 0;
    if (fy_maybe_zero_mod(D)) zero_mod = fy_is_zero_mod(D) ? 1 : 0;
    const bool same_x = q4_perm_i32<0, 0, 0, 0>(zero_mod) != 0 && !idA && !idB;
    const bool same_y = q4_perm_i32<1, 1, 1, 1>(zero_mod) != 0;
    const Fy<F> E = fy_sqr(q4_perm<F, 0, 1, 0, 0>(D));
    const Fy<F> F4 = fy_mul(q4_select((q & 1) != 0, q4_perm<F, 0, 0, 0, 0>(D), T1), q4_perm<F, 0, 0, 2, 3>(E));
    const Fy<F> X3 = fy_sub_sub2(q4_perm<F, 1, 1, 1, 1>(E), q4_perm<F, 1, 1, 1, 1>(F4), q4_perm<F, 0, 0, 0, 0>(F4));
    const Fy<F> a5 = q4_select(q == 1, D, q4_perm<F, 1, 1, 2, 3>(T1));
    const Fy<F> b5 = q4_select(q == 1, fy_sub_lazy(q4_perm<F, 0, 0, 0, 0>(F4), X3), q4_perm<F, 1, 1, 2, 3>(F4));
    const Fy<F> G = fy_mul(a5, b5);
    const Fy<F> Y3 = fy_sub(G, q4_perm<F, 0, 0, 0, 0>(G));
    Fy<F> r = q4_select(q == 0, X3, q4_select(q == 1, Y3, q4_select(q == 2, F4, G)));
    if (__any(same_x ? 1 : 0)) {
        const Fy<F> dbl = q4_dbl(A, q);
        r = q4_select(same_x, q4_select(same_y, dbl, fy_zero<F>()), r);
    }
    return q4_select(idA, B, q4_select(idB, A, r));
}

// coordinate q of a raw point (curve.h XYZZzMem: 36 limbs, x y zz zzz) and back
template <class F>
__device__ __forceinline__ Fy<F> q4_load(const XYZZzMem* src, const int q) {
    const u32* w = src->w + NLIMBS * q;
    Fy<F> r;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = (i32)w[i];
    return q4_select(q4_is_identity(r), fy_zero<F>(), r);  // an identity is stored with zz = 0 only: the other lanes' limbs are not to be read
}
template <class F>
__device__ __forceinline__ void q4_store(XYZZzMem* dst, const int q, const Fy<F>& v) {
    u32* w = dst->w + NLIMBS * q;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) w[i] = (u32)v.l[i];
}
#endif  // __HIPCC__

}  // namespace trh
