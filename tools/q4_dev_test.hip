// device vs CPU model of csrc/curve_q4.h's q4_add, intermediate by intermediate (debugging aid; tools/, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../tiny-ram-halo2_amd/csrc/curve_q4.h"
using namespace trh;
typedef FpParams F;
struct Dump { Fy<F> v[10][4]; };  // T1 T2 D E F4 X3 G Y3 R(final) spare
__global__ void k(const Fy<F>* in, Dump* out) {
    const int q = threadIdx.x & 3, pair = threadIdx.x >> 2;
    const Fy<F> A = in[pair * 8 + q], B = in[pair * 8 + 4 + q];
    Dump& d = out[pair];
    const Fy<F> T1 = fy_mul(A, q4_perm<F, 2, 3, 2, 3>(B));
    const Fy<F> T2 = fy_mul(B, q4_perm<F, 2, 3, 2, 3>(A));
    const Fy<F> D = fy_sub(T2, T1);
    const Fy<F> E = fy_sqr(q4_perm<F, 0, 1, 0, 0>(D));
    const Fy<F> F4 = fy_mul(q4_select((q & 1) != 0, q4_perm<F, 0, 0, 0, 0>(D), T1), q4_perm<F, 0, 0, 2, 3>(E));
    const Fy<F> X3 = fy_sub_sub2(q4_perm<F, 1, 1, 1, 1>(E), q4_perm<F, 1, 1, 1, 1>(F4), q4_perm<F, 0, 0, 0, 0>(F4));
    const Fy<F> a5 = q4_select(q == 1, D, q4_perm<F, 1, 1, 2, 3>(T1));
    const Fy<F> b5 = q4_select(q == 1, fy_sub_lazy(q4_perm<F, 0, 0, 0, 0>(F4), X3), q4_perm<F, 1, 1, 2, 3>(F4));
    const Fy<F> G = fy_mul(a5, b5);
    const Fy<F> Y3 = fy_sub(G, q4_perm<F, 0, 0, 0, 0>(G));
    d.v[0][q] = T1; d.v[1][q] = T2; d.v[2][q] = D; d.v[3][q] = E; d.v[4][q] = F4; d.v[5][q] = X3; d.v[6][q] = G; d.v[7][q] = Y3;
    d.v[8][q] = q4_add(A, B, q); d.v[9][q] = a5;
}
static u64 seed = 0x243f6a8885a308d3ull;
static u64 nxt() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; }
static void perm(const Fy<F>* v, int p0, int p1, int p2, int p3, Fy<F>* r) { r[0] = v[p0]; r[1] = v[p1]; r[2] = v[p2]; r[3] = v[p3]; }
int main() {
    const int pairs = 16;
    Fy<F> in[pairs * 8];
    Affine<F> G; G.x = fe_neg(fe_one<F>()); G.y = fe_dbl(fe_one<F>());
    XYZZz<F> a = xyzzz_from_canonical(xyzz_from_affine(G)), b = xyzzz_dbl(a);
    for (int p = 0; p < pairs; ++p) {
        in[p * 8 + 0] = a.x; in[p * 8 + 1] = a.y; in[p * 8 + 2] = a.zz; in[p * 8 + 3] = a.zzz;
        in[p * 8 + 4] = b.x; in[p * 8 + 5] = b.y; in[p * 8 + 6] = b.zz; in[p * 8 + 7] = b.zzz;
        const XYZZz<F> c = xyzzz_add(a, b); a = xyzzz_dbl(b); b = c;
    }
    Fy<F>* din; Dump* dout; Dump h[pairs];
    (void)hipMalloc(&din, sizeof(in)); (void)hipMalloc(&dout, sizeof(h));
    (void)hipMemcpy(din, in, sizeof(in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(pairs * 4), 0, 0, din, dout);
    (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[10] = {"T1", "T2", "D", "E", "F4", "X3", "G", "Y3", "q4_add", "a5"};
    int bad = 0;
    for (int p = 0; p < pairs; ++p) {
        const Fy<F>* A = &in[p * 8]; const Fy<F>* B = A + 4;
        Fy<F> m[10][4], t[4], u[4];
        perm(B, 2, 3, 2, 3, t); for (int q = 0; q < 4; ++q) m[0][q] = fy_mul(A[q], t[q]);
        perm(A, 2, 3, 2, 3, t); for (int q = 0; q < 4; ++q) { m[1][q] = fy_mul(B[q], t[q]); m[2][q] = fy_sub(m[1][q], m[0][q]); }
        perm(m[2], 0, 1, 0, 0, t); for (int q = 0; q < 4; ++q) m[3][q] = fy_sqr(t[q]);
        perm(m[2], 0, 0, 0, 0, t); perm(m[3], 0, 0, 2, 3, u); for (int q = 0; q < 4; ++q) m[4][q] = fy_mul((q & 1) ? t[q] : m[0][q], u[q]);
        Fy<F> e1[4], f1[4], f0[4]; perm(m[3], 1, 1, 1, 1, e1); perm(m[4], 1, 1, 1, 1, f1); perm(m[4], 0, 0, 0, 0, f0);
        for (int q = 0; q < 4; ++q) m[5][q] = fy_sub_sub2(e1[q], f1[q], f0[q]);
        Fy<F> t1p[4], f4p[4]; perm(m[0], 1, 1, 2, 3, t1p); perm(m[4], 1, 1, 2, 3, f4p);
        for (int q = 0; q < 4; ++q) { m[9][q] = q == 1 ? m[2][q] : t1p[q]; m[6][q] = fy_mul(m[9][q], q == 1 ? fy_sub_lazy(f0[q], m[5][q]) : f4p[q]); }
        perm(m[6], 0, 0, 0, 0, t); for (int q = 0; q < 4; ++q) m[7][q] = fy_sub(m[6][q], t[q]);
        for (int q = 0; q < 4; ++q) m[8][q] = q == 0 ? m[5][q] : q == 1 ? m[7][q] : q == 2 ? m[4][q] : m[6][q];
        for (int s = 0; s < 10; ++s) for (int q = 0; q < 4; ++q)
            if (memcmp(&m[s][q], &h[p].v[s][q], sizeof(Fy<F>)) != 0) { if (++bad <= 12) printf("pair %d %s lane %d differs: dev l0 %d l8 %d, model l0 %d l8 %d\n", p, names[s], q, h[p].v[s][q].l[0], h[p].v[s][q].l[8], m[s][q].l[0], m[s][q].l[8]); }
    }
    printf(bad ? "q4 device vs model: %d differences\n" : "q4 device vs model: identical\n", bad);
    return 0;
}
