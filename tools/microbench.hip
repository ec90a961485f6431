// Instruction- and field-level microbenchmarks on gfx950 (secondary diagnostic of SURVEY.md 8d:
// both hot kernels are bound by 32-bit integer multiply issue, not by HBM).
// Build: hipcc -O3 --offload-arch=gfx950 -I tiny-ram-halo2_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "curve.h"
using namespace trh;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;

__global__ void k_mad64(u64* out, u32 a, u32 b) {
    u64 x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    u32 aa = a + threadIdx.x, bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        x0 = (u64)aa * bb + x0; x1 = (u64)aa * bb + x1; x2 = (u64)aa * bb + x2; x3 = (u64)aa * bb + x3;
        x4 = (u64)aa * bb + x4; x5 = (u64)aa * bb + x5; x6 = (u64)aa * bb + x6; x7 = (u64)aa * bb + x7;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_mullo(u64* out, u32 a, u32 b) {
    u32 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    u32 bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = x[k] * bb;
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u32 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_mulhi(u64* out, u32 a, u32 b) {
    u32 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k + 0x80000000u;
    u32 bb = b + threadIdx.x + 0xf0000000u;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __umulhi(x[k], bb) + 0x80000000u;
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u32 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_add32(u64* out, u32 a, u32 b) {
    u32 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    u32 bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = x[k] + bb;
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u32 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_addc(u64* out, u32 a, u32 b) {  // 8-limb carry chain (v_add_co + 7 v_addc_co)
    u32 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    u32 bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        u32 c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __builtin_addc(x[k], bb, c, &c);
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u32 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_fma64(u64* out, u32 a, u32 b) {
    double x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    double aa = 1.0000001 + a * 1e-9, bb = b * 1e-9;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __builtin_fma(x[k], aa, bb);
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    double r = 0;
    for (int k = 0; k < 8; ++k) r += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)r;
}
__global__ void k_mad24(u64* out, u32 a, u32 b) {
    u32 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    u32 bb = (b + threadIdx.x) & 0xffffff;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __umul24(x[k], bb) + 7u;
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u32 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void k_add64(u64* out, u32 a, u32 b) {  // 64-bit add (v_lshl_add_u64 / add_co+addc)
    u64 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    u64 bb = ((u64)b << 33) + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = x[k] + bb;
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_shr64(u64* out, u32 a, u32 b) {  // v_lshrrev_b64 by a constant
    u64 x[8];
    for (int k = 0; k < 8; ++k) x[k] = ((u64)threadIdx.x << 40) + k + ((u64)a << 50);
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (x[k] >> 30) | 0x8000000000000000ull;
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_alignbit(u64* out, u32 a, u32 b) {
    u32 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    u32 bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __builtin_amdgcn_alignbit(bb, x[k], 30);
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u32 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_add3(u64* out, u32 a, u32 b) {
    u32 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    u32 bb = b + threadIdx.x, cc = a ^ threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = x[k] + bb + cc;
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    u32 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void k_addco(u64* out, u32 a, u32 b) {  // independent add_co + addc pairs (no long chain)
    u32 x[8], y[8];
    for (int k = 0; k < 8; ++k) { x[k] = threadIdx.x + k; y[k] = k; }
    u32 bb = b + threadIdx.x + 0xfff00000u;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { u32 c = 0; x[k] = __builtin_addc(x[k], bb, 0u, &c); y[k] = __builtin_addc(y[k], 0u, c, &c); }
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
    }
    u32 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x[k] ^ y[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

constexpr int FITERS = 512;
template <int MODE>
__global__ void __launch_bounds__(256) k_field(u64* out, u32 seed) {
    Fe<FpParams> x = fe_one<FpParams>(), y = fe_r2<FpParams>();
    x.l[0] += threadIdx.x + seed; y.l[1] ^= blockIdx.x;
    for (int i = 0; i < FITERS; ++i) {
        if (MODE == 0) { const Fe<FpParams> t = fe_mul(x, y); x = y; y = t; }  // both operands vary
        else if (MODE == 1) x = fe_sqr(x);
        else if (MODE == 2) x = fe_add(x, y);
        else x = fe_sub(x, y);
    }
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r ^= x.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_lazy(u64* out, u32 seed) {
    Fy<FpParams> x = fy_one<FpParams>(), y = fy_one<FpParams>();
    x.l[0] += threadIdx.x + seed; y.l[1] ^= blockIdx.x;
    for (int i = 0; i < FITERS; ++i) {
        if (MODE == 0) { const Fy<FpParams> t = fy_mul(x, y); x = y; y = t; }  // both operands vary
        else if (MODE == 1) x = fy_sqr(x);
        else if (MODE == 2) x = fy_add(x, y);
        else x = fy_sub(x, y);
        if (MODE >= 2) { x.l[8] &= 0xffff; }
    }
    u64 r = 0;
    for (int k = 0; k < 9; ++k) r ^= x.l[k] ^ y.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_madd_lazy(u64* out, u32 seed) {
    AffineZ<FpParams> g;
    g.x = fy_one<FpParams>(); g.y = fy_one<FpParams>(); g.x.l[0] += 5; g.y.l[1] += 7;   // not a curve point: timing only
    XYZZz<FpParams> acc;
    acc.x = g.y; acc.y = g.x; acc.zz = fy_one<FpParams>(); acc.zzz = fy_one<FpParams>();
    acc.x.l[2] += threadIdx.x + seed;
    for (int i = 0; i < FITERS; ++i) xyzzz_madd(acc, g);
    u64 r = 0;
    for (int k = 0; k < 9; ++k) r ^= acc.x.l[k] ^ acc.zz.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_madd(u64* out, u32 seed) {
    Affine<FpParams> g;
    g.x = fe_neg(fe_one<FpParams>()); g.y = fe_dbl(fe_one<FpParams>());
    XYZZ<FpParams> acc = xyzz_dbl_affine(g);
    for (int i = 0; i < (int)(threadIdx.x & 3) + (int)(seed & 1); ++i) acc = xyzz_dbl(acc);
    for (int i = 0; i < FITERS; ++i) xyzz_madd(acc, g);
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r ^= acc.x.l[k] ^ acc.zz.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class K>
float run(K kern, int blocks, int threads, u64* d_out, const char* name, double ops_per_thread) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern(blocks, threads, d_out);  // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) kern(blocks, threads, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    double total = ops_per_thread * blocks * threads;
    printf("%-28s blocks=%5d thr=%4d  %8.3f ms  %10.2f Gop/s\n", name, blocks, threads, ms, total / ms * 1e-6);
    return ms;
}

int main() {
    u64* d_out;
    CK(hipMalloc(&d_out, sizeof(u64) * 4096 * 1024));
    const int B = 256 * 8, T = 256;  // 8 blocks of 4 waves per CU = 8 waves / SIMD
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_mad64, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_mad_u64_u32", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_mullo, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_mul_lo_u32", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_mulhi, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_mul_hi_u32 (+add)", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_add32, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_add_u32", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_addc, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_addc_co chain (per limb)", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_fma64, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_fma_f64", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_mad24, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_mul_u32_u24 (+add)", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_add64, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "64-bit add", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_shr64, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "64-bit shr 30 (+or)", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_alignbit, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_alignbit_b32", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_add3, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "v_add3_u32", 8.0 * ITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_addco, dim3(b), dim3(t), 0, 0, o, 3u, 5u); }, B, T, d_out, "add_co+addc pair (x2 ops)", 16.0 * ITERS);
    for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
        run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_field<0>, dim3(b), dim3(t), 0, 0, o, 1u); }, blocks, T, d_out, "fe_mul", FITERS);
        run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_field<1>, dim3(b), dim3(t), 0, 0, o, 1u); }, blocks, T, d_out, "fe_sqr", FITERS);
    }
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_field<2>, dim3(b), dim3(t), 0, 0, o, 1u); }, B, T, d_out, "fe_add", FITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_field<3>, dim3(b), dim3(t), 0, 0, o, 1u); }, B, T, d_out, "fe_sub", FITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_lazy<0>, dim3(b), dim3(t), 0, 0, o, 1u); }, B, T, d_out, "fy_mul (signed lazy R''=2^261)", FITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_lazy<1>, dim3(b), dim3(t), 0, 0, o, 1u); }, B, T, d_out, "fy_sqr", FITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_lazy<2>, dim3(b), dim3(t), 0, 0, o, 1u); }, B, T, d_out, "fy_add", FITERS);
    run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_lazy<3>, dim3(b), dim3(t), 0, 0, o, 1u); }, B, T, d_out, "fy_sub", FITERS);
    for (int blocks : {256 * 3, 256 * 6})
        run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_madd_lazy, dim3(b), dim3(t), 0, 0, o, 1u); }, blocks, T, d_out, "xyzzz_madd (lazy)", FITERS);
    for (int blocks : {256 * 2, 256 * 4, 256 * 8})
        run([](int b, int t, u64* o) { hipLaunchKernelGGL(k_madd, dim3(b), dim3(t), 0, 0, o, 1u); }, blocks, T, d_out, "xyzz_madd", FITERS);
    return 0;
}
