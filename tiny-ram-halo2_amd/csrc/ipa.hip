// Device primitives of the inner-product-argument opening (halo2_proofs 0.2.0
// `poly::commitment::prover::create_proof`, reached through `poly::multiopen::create_proof` inside
// plonk::create_proof -- reference call site /root/reference/src/test_utils.rs:41-49; SURVEY.md
// section 8 row a7).  Per round j (half = 2^(k-j-1)) the Rust code does
//     L_j = <p'[half..], G'[..half]>          R_j = <p'[..half], G'[half..]>          (two MSMs)
//     value_l = <p'[half..], b[..half]>       value_r = <p'[..half], b[half..]>       (inner products)
//     p'[i] += u^-1 p'[i+half]      b[i] += u b[i+half]                               (folds)
//     G'[i] = G'[i] + u G'[i+half]   then batch_normalize                             (generator collapse)
// The MSMs run through msm.hip on the live G' buffer (trh_bases_wrap_device + offset); this file
// holds the rest: compute_inner_product, the folds (axpy), the generator collapse and the
// power vector b = (1, x, x^2, ...).
#include <string.h>

#include "ctx.h"

namespace trh {
namespace {

template <class F>
__device__ __forceinline__ Fe<F> ld(const uint4* p) {
    uint4 a = p[0], b = p[1];
    return fe_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
template <class F>
__device__ __forceinline__ void st(uint4* p, const Fe<F>& v) {
    u32 w[8];
    fe_store(v, w);
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// partial[block] = sum over the block's grid-stride share of a[i] * b[i]
template <class F>
__global__ void __launch_bounds__(256) inner_product_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t n, uint4* __restrict__ partial) {
    __shared__ Fe<F> sh[256];
    Fe<F> acc = fe_zero<F>();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc = fe_add(acc, fe_mul(ld<F>(a + 2 * i), ld<F>(b + 2 * i)));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st<F>(partial + 2 * blockIdx.x, sh[0]);
}
template <class F>
__global__ void __launch_bounds__(256) sum_partials_kernel(const uint4* __restrict__ partial, u32 count, uint4* __restrict__ out) {
    __shared__ Fe<F> sh[256];
    Fe<F> acc = fe_zero<F>();
    for (u32 i = threadIdx.x; i < count; i += 256) acc = fe_add(acc, ld<F>(partial + 2 * i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st<F>(out, sh[0]);
}

// y[i] += c * x[i]
template <class F>
__global__ void __launch_bounds__(256) axpy_kernel(uint4* __restrict__ y, const uint4* __restrict__ x, size_t n, const uint4* __restrict__ c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st<F>(y + 2 * i, fe_add(ld<F>(y + 2 * i), fe_mul(ld<F>(x + 2 * i), ld<F>(c))));
}

// out[i] = x^i, pw[b] = x^(2^b)
template <class F>
__global__ void __launch_bounds__(256) powers_kernel(uint4* __restrict__ out, size_t n, const uint4* __restrict__ pw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> r = fe_one<F>();
    for (int b = 0; b < 32; ++b)
        if ((i >> b) & 1u) r = fe_mul(r, ld<F>(pw + 2 * b));
    st<F>(out + 2 * i, r);
}

// generator collapse: g_lo[i] = g_lo[i] + u * g_hi[i], normalised to affine
// (u given as canonical 32-bit words; double-and-add from the top set bit)
template <class BF>
__global__ void __launch_bounds__(256) bases_fold_kernel(AffineMem* __restrict__ g_lo, const AffineMem* __restrict__ g_hi, size_t half, const u32* __restrict__ u_words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const Affine<BF> hi = aff_load<BF>(g_hi[i]);
    XYZZ<BF> acc = xyzz_identity<BF>();
    for (int w = 7; w >= 0; --w) {
        const u32 word = u_words[w];  // wave-uniform
        for (int bit = 31; bit >= 0; --bit) {
            acc = xyzz_dbl(acc);
            if ((word >> bit) & 1u) xyzz_madd(acc, hi);
        }
    }
    xyzz_madd(acc, aff_load<BF>(g_lo[i]));
    aff_store(xyzz_to_affine(acc), g_lo[i]);
}

template <class F>
int inner_product_t(const void* a, const void* b, size_t n, hipStream_t s, u64* out) {
    Ctx& c = ctx();
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    TRH_TRY(c.io.ensure((size_t)(blocks + 1) * 32));
    uint4* partial = c.io.as<uint4>();
    uint4* result = partial + 2 * blocks;
    hipLaunchKernelGGL((inner_product_kernel<F>), dim3(blocks), dim3(256), 0, s, (const uint4*)a, (const uint4*)b, n, partial);
    hipLaunchKernelGGL((sum_partials_kernel<F>), dim3(1), dim3(256), 0, s, partial, blocks, result);
    TRH_HIP_TRY(hipGetLastError());
    TRH_HIP_TRY(hipMemcpyAsync(out, result, 32, hipMemcpyDeviceToHost, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));
    return TRH_OK;
}

// small per-call constant staged in the factor ring (see trh_field_scale_rows_dev)
int stage_constant(const void* host, size_t bytes, hipStream_t s, void** dev) {
    Ctx& c = ctx();
    TRH_TRY(c.factors.ensure(16 * 64 * 32));
    char* slot = (char*)c.factors.p + (size_t)(c.factor_slot++ & 15) * 64 * 32;
    TRH_HIP_TRY(hipMemcpyAsync(slot, host, bytes, hipMemcpyHostToDevice, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));
    *dev = slot;
    return TRH_OK;
}

template <class F>
int powers_t(void* out, size_t n, const u64* x, hipStream_t s) {
    FeMem pw[32];
    memcpy(&pw[0], x, 32);
    for (int b = 1; b < 32; ++b) fe_store(fe_sqr(fe_load<F>(pw[b - 1])), pw[b]);
    void* d_pw;
    TRH_TRY(stage_constant(pw, sizeof(pw), s, &d_pw));
    hipLaunchKernelGGL((powers_kernel<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4*)out, n, (const uint4*)d_pw);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

template <class SF, class BF>
int bases_fold_t(void* g_lo, const void* g_hi, size_t half, const u64* u_mont, hipStream_t s) {
    FeMem um, uc;
    memcpy(&um, u_mont, 32);
    fe_store(fe_from_mont(fe_load<SF>(um)), uc);  // scalar -> canonical bits (host)
    void* d_u;
    TRH_TRY(stage_constant(&uc, 32, s, &d_u));
    hipLaunchKernelGGL((bases_fold_kernel<BF>), dim3((unsigned)((half + 255) / 256)), dim3(256), 0, s, (AffineMem*)g_lo, (const AffineMem*)g_hi, half, (const u32*)d_u);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

}  // namespace
}  // namespace trh

using namespace trh;

extern "C" {

int trh_field_inner_product_dev(int field, const void* a_dev, const void* b_dev, size_t n, void* stream, uint64_t out[4]) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!out || (n && (!a_dev || !b_dev))) { set_error("inner_product: null pointer"); return TRH_EINVAL; }
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (field == TRH_FP) return inner_product_t<FpParams>(a_dev, b_dev, n, (hipStream_t)stream, out);
    return inner_product_t<FqParams>(a_dev, b_dev, n, (hipStream_t)stream, out);
}

int trh_field_axpy_dev(int field, void* y_dev, const void* x_dev, size_t n, const uint64_t c_mont[4], void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!c_mont || (n && (!y_dev || !x_dev))) { set_error("axpy: null pointer"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    void* d_c;
    TRH_TRY(stage_constant(c_mont, 32, (hipStream_t)stream, &d_c));
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (field == TRH_FP) hipLaunchKernelGGL((axpy_kernel<FpParams>), dim3(gb), dim3(256), 0, (hipStream_t)stream, (uint4*)y_dev, (const uint4*)x_dev, n, (const uint4*)d_c);
    else hipLaunchKernelGGL((axpy_kernel<FqParams>), dim3(gb), dim3(256), 0, (hipStream_t)stream, (uint4*)y_dev, (const uint4*)x_dev, n, (const uint4*)d_c);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

int trh_field_powers_dev(int field, void* out_dev, size_t n, const uint64_t x_mont[4], void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!x_mont || (n && !out_dev)) { set_error("powers: null pointer"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (field == TRH_FP) return powers_t<FpParams>(out_dev, n, x_mont, (hipStream_t)stream);
    return powers_t<FqParams>(out_dev, n, x_mont, (hipStream_t)stream);
}

int trh_bases_fold_dev(int curve, void* g_lo_dev, const void* g_hi_dev, size_t half, const uint64_t u_mont[4], void* stream) {
    TRH_TRY(require_init());
    if (curve != TRH_PALLAS && curve != TRH_VESTA) { set_error("unknown curve id %d", curve); return TRH_EINVAL; }
    if (!u_mont || (half && (!g_lo_dev || !g_hi_dev))) { set_error("bases_fold: null pointer"); return TRH_EINVAL; }
    if (!half) return TRH_OK;
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    // pallas: scalar field Fq, base field Fp
    if (curve == TRH_PALLAS) return bases_fold_t<FqParams, FpParams>(g_lo_dev, g_hi_dev, half, u_mont, (hipStream_t)stream);
    return bases_fold_t<FpParams, FqParams>(g_lo_dev, g_hi_dev, half, u_mont, (hipStream_t)stream);
}

}  // extern "C"
