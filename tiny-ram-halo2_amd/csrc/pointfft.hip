// Radix-2 FFT over curve points: halo2_proofs 0.2.0 `arithmetic::best_fft::<C::Curve>` as
// `poly::commitment::Params::new(k)` uses it to turn the generators g into the Lagrange-basis
// generators g_lagrange (reference call site /root/reference/src/test_utils.rs:21, 89; SURVEY.md
// section 8 rows a4 "G = field element or curve point" and f-3):
//     a'[i] = sum_j [omega^(i j)] a[j],      then every point times n^-1, then batch_normalize.
// Same contract as the field transform: in place, natural order in and out.  Each butterfly is one
// 255-bit scalar multiplication (double-and-add on XYZZ accumulators) plus an add and a subtract, so
// the transform is n/2 * log n scalar multiplications -- a one-off setup cost per Params.
#include <string.h>

#include "ctx.h"

namespace trh {
namespace {

template <class F>
__device__ __forceinline__ XYZZ<F> ld_xyzz(const XYZZMem* p) { return xyzz_load<F>(*p); }

// work[bitrev(i)] = points[i] as XYZZ
template <class BF>
__global__ void __launch_bounds__(256) pfft_load_kernel(const AffineMem* __restrict__ pts, XYZZMem* __restrict__ work, u32 log_n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    const u32 r = log_n ? (__brev(i) >> (32 - log_n)) : 0u;
    xyzz_store(xyzz_from_affine(aff_load<BF>(pts[i])), work[r]);
}

// tw[j] = omega^j as canonical 32-bit words (scalar bits for double-and-add); pw[b] = omega^(2^b), Montgomery
template <class SF>
__global__ void __launch_bounds__(256) pfft_twiddle_kernel(FeMem* __restrict__ tw, u32 count, const FeMem* __restrict__ pw) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fe<SF> r = fe_one<SF>();
    for (int b = 0; b < 32; ++b)
        if ((j >> b) & 1u) r = fe_mul(r, fe_load<SF>(pw[b]));
    fe_store(fe_from_mont(r), tw[j]);
}

// [k] p, k given as 8 canonical words in memory
template <class BF>
__device__ XYZZ<BF> scalar_mul(const XYZZ<BF>& p, const u32* __restrict__ k) {
    XYZZ<BF> acc = xyzz_identity<BF>();
    for (int w = 7; w >= 0; --w) {
        const u32 word = k[w];
        for (int bit = 31; bit >= 0; --bit) {
            acc = xyzz_dbl(acc);
            if ((word >> bit) & 1u) acc = xyzz_add(acc, p);
        }
    }
    return acc;
}

template <class BF>
__global__ void __launch_bounds__(256) pfft_stage_kernel(XYZZMem* __restrict__ work, const FeMem* __restrict__ tw, u32 log_n, u32 st) {
    const u32 bf = blockIdx.x * blockDim.x + threadIdx.x;
    if (bf >= (1u << (log_n - 1))) return;
    const u32 half = 1u << st, pos = bf & (half - 1u);
    const u32 i0 = ((bf >> st) << (st + 1)) | pos, i1 = i0 + half;
    const XYZZ<BF> a = ld_xyzz<BF>(&work[i0]);
    XYZZ<BF> t = ld_xyzz<BF>(&work[i1]);
    if (pos) t = scalar_mul<BF>(t, tw[pos << (log_n - 1 - st)].w);
    xyzz_store(xyzz_add(a, t), work[i0]);
    xyzz_store(xyzz_add(a, xyzz_neg(t)), work[i1]);
}

template <class BF>
__global__ void __launch_bounds__(256) pfft_finish_kernel(const XYZZMem* __restrict__ work, AffineMem* __restrict__ pts, u32 n, const FeMem* __restrict__ scale) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<BF> p = ld_xyzz<BF>(&work[i]);
    if (scale) p = scalar_mul<BF>(p, scale->w);
    aff_store(xyzz_to_affine(p), pts[i]);
}

template <class SF, class BF>
int point_fft_t(void* points_dev, uint32_t log_n, const u64* omega, const u64* scale, hipStream_t s) {
    Ctx& c = ctx();
    const size_t n = (size_t)1 << log_n;
    const u32 ntw = log_n ? (u32)(n >> 1) : 1u;
    // scratch: XYZZ work array, twiddle scalars, omega^(2^b) table, optional scale
    TRH_TRY(c.pfft.ensure(n * sizeof(XYZZMem) + (size_t)ntw * 32 + 34 * 32));
    XYZZMem* work = c.pfft.as<XYZZMem>();
    FeMem* tw = (FeMem*)(work + n);
    FeMem* pw = tw + ntw;
    FeMem* d_scale = pw + 32;
    FeMem host[33];
    memcpy(&host[0], omega, 32);
    for (int b = 1; b < 32; ++b) fe_store(fe_sqr(fe_load<SF>(host[b - 1])), host[b]);
    if (scale) {
        FeMem sm;
        memcpy(&sm, scale, 32);
        fe_store(fe_from_mont(fe_load<SF>(sm)), host[32]);
    }
    TRH_HIP_TRY(hipMemcpyAsync(pw, host, sizeof(host), hipMemcpyHostToDevice, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));  // host[] is a stack buffer
    hipLaunchKernelGGL((pfft_twiddle_kernel<SF>), dim3((ntw + 255) / 256), dim3(256), 0, s, tw, ntw, pw);
    hipLaunchKernelGGL((pfft_load_kernel<BF>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const AffineMem*)points_dev, work, log_n);
    for (u32 st = 0; st < log_n; ++st)
        hipLaunchKernelGGL((pfft_stage_kernel<BF>), dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, work, tw, log_n, st);
    hipLaunchKernelGGL((pfft_finish_kernel<BF>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, work, (AffineMem*)points_dev, (u32)n, scale ? d_scale : nullptr);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

}  // namespace
}  // namespace trh

using namespace trh;

extern "C" int trh_point_fft_dev(int curve, void* points_dev, uint32_t log_n, const uint64_t omega[4], const uint64_t* scale_or_null, void* stream) {
    TRH_TRY(require_init());
    if (curve != TRH_PALLAS && curve != TRH_VESTA) { set_error("unknown curve id %d", curve); return TRH_EINVAL; }
    if (!points_dev || !omega) { set_error("point_fft: null pointer"); return TRH_EINVAL; }
    if (log_n > 24) { set_error("point_fft: log_n %u > 24 unsupported", log_n); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_point_fft_dev");
    Ctx& c = ctx();
    (void)c;
    // pallas: scalar field Fq, base field Fp
    if (curve == TRH_PALLAS) return point_fft_t<FqParams, FpParams>(points_dev, log_n, omega, scale_or_null, (hipStream_t)stream);
    return point_fft_t<FpParams, FqParams>(points_dev, log_n, omega, scale_or_null, (hipStream_t)stream);
}
