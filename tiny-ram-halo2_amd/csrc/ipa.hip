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

#include <chrono>

#include "ctx.h"
#include "hostcombine.h"
#include "hosthelper.h"

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

// evaluation of a batch of polynomials at one point: partial[poly][block] = sum over the block's share of a[poly][i] * pw[i]
template <class F>
__global__ void __launch_bounds__(256) eval_batch_kernel(const uint4* __restrict__ polys, const uint4* __restrict__ pw, size_t n, uint4* __restrict__ partial) {
    __shared__ Fe<F> sh[256];
    const uint4* a = polys + (size_t)blockIdx.y * n * 2;
    Fe<F> acc = fe_zero<F>();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc = fe_add(acc, fe_mul(ld<F>(a + 2 * i), ld<F>(pw + 2 * i)));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st<F>(partial + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x), sh[0]);
}
template <class F>
__global__ void __launch_bounds__(256) sum_partials_batch_kernel(const uint4* __restrict__ partial, u32 count, uint4* __restrict__ out) {
    __shared__ Fe<F> sh[256];
    const uint4* p = partial + 2 * (size_t)blockIdx.x * count;
    Fe<F> acc = fe_zero<F>();
    for (u32 i = threadIdx.x; i < count; i += 256) acc = fe_add(acc, ld<F>(p + 2 * i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st<F>(out + 2 * (size_t)blockIdx.x, sh[0]);
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

// ---- generator collapse without point arithmetic (native prover) ---------------------------
// The folded generators are linear combinations of the original ones,
//     G'_j[i] = sum over idx = i (mod n / 2^j) of wgt[idx] * G[idx],
// and a collapse with challenge u multiplies wgt[idx] by u on every index whose bit log2(half) is set.
// So L_j = <p'[half..], G'[..half]> and R_j = <p'[..half], G'[half..]> are MSMs over the ORIGINAL
// resident bases with the scalars below (zero outside their half), and G' is never materialised.
// a round's constants travel as kernel arguments (a staged copy is a blit kernel and ~20 us of queue idle per round on the trace)
struct IpaConsts { uint4 w[6]; };
template <class F>
__device__ __forceinline__ Fe<F> ipa_const(const IpaConsts& c, int k) { return fe_load<F>(c.w[2 * k].x, c.w[2 * k].y, c.w[2 * k].z, c.w[2 * k].w, c.w[2 * k + 1].x, c.w[2 * k + 1].y, c.w[2 * k + 1].z, c.w[2 * k + 1].w); }
// the folds of a round in one launch: p'[i] += u^-1 p'[i + half], b[i] += u b[i + half] (i < half) and the generators' weights
// (x u where bit log2(half) of the index is set); consts = (u^-1, u)
template <class F>
__global__ void __launch_bounds__(256) ipa_round_update_kernel(uint4* __restrict__ p, uint4* __restrict__ b, uint4* __restrict__ wgt, size_t n, size_t half, u32 bit,
                                                               const IpaConsts consts) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const Fe<F> u = ipa_const<F>(consts, 1);
    if ((idx >> bit) & 1u) st<F>(wgt + 2 * idx, fe_mul(ld<F>(wgt + 2 * idx), u));
    if (idx < half) {
        st<F>(p + 2 * idx, fe_add(ld<F>(p + 2 * idx), fe_mul(ld<F>(p + 2 * (half + idx)), ipa_const<F>(consts, 0))));
        st<F>(b + 2 * idx, fe_add(ld<F>(b + 2 * idx), fe_mul(ld<F>(b + 2 * (half + idx)), u)));
    }
}

// The folds of round j - 1 and the scalar rows of round j in ONE launch (round 6; two launches before):
//   * the folds are applied ON THE FLY: p'_new[x] = p'[x] + u^-1 p'[hprev + x], b_new[x] = b[x] + u b[hprev + x] (x < hprev) are computed
//     where this round's scalar rows read them and written once to the OTHER buffer of a ping-pong pair (a thread's reads are other
//     threads' writes); the weights are updated in place (own index);
//   * scalars of L_j (row 0) and R_j (row 1) over the bases g || w || u: v = wgt[idx] p'_new[...] on the half of the indices the row
//     covers, 0 on the other.
// consts = (u_prev^-1, u_prev); first = 1 in round 0 (nothing to fold yet: p' and b are read as they are and stay where they are).
// (Measured and dropped: the two inner products and the tail scalars in the same launch behind a "last block" ticket -- every block's release
//  fence and the last block's acquire made it 33 - 67 us against 29 us for the four launches of round 5.)
// both inner products of an IPA round in one launch: partial[pair][block]
template <class F>
__global__ void __launch_bounds__(256) inner_product2_kernel(const uint4* __restrict__ a0, const uint4* __restrict__ b0, const uint4* __restrict__ a1, const uint4* __restrict__ b1, size_t n,
                                                             uint4* __restrict__ partial) {
    __shared__ Fe<F> sh[256];
    const uint4* a = blockIdx.y ? a1 : a0;
    const uint4* b = blockIdx.y ? b1 : b0;
    Fe<F> acc = fe_zero<F>();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc = fe_add(acc, fe_mul(ld<F>(a + 2 * i), ld<F>(b + 2 * i)));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st<F>(partial + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x), sh[0]);
}

template <class F>
__global__ void __launch_bounds__(256) ipa_round_front_kernel(const uint4* __restrict__ p_old, const uint4* __restrict__ b_old, uint4* __restrict__ p_new, uint4* __restrict__ b_new,
                                                              uint4* __restrict__ wgt, uint4* __restrict__ lrsc, size_t n, size_t half, u32 bit, size_t stride, int first,
                                                              int wfresh /* the weights are all one and not in memory yet (rounds 0 and 1) */,
                                                              int canon /* the scalar rows leave in canonical form (the small-MSM kernel would convert every scalar in each of its 104 workgroups) */,
                                                              const IpaConsts consts) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const size_t hprev = half << 1;  // the previous round's half
    const Fe<F> uinv = ipa_const<F>(consts, 0), u = ipa_const<F>(consts, 1);
    auto pn = [&](size_t x) { return first ? ld<F>(p_old + 2 * x) : fe_add(ld<F>(p_old + 2 * x), fe_mul(ld<F>(p_old + 2 * (hprev + x)), uinv)); };  // x < hprev
    Fe<F> w;
    if (wfresh) {  // round 0 reads no weights, round 1 writes them all
        w = (!first && ((idx >> (bit + 1)) & 1u)) ? u : fe_one<F>();
        if (!first) st<F>(wgt + 2 * idx, w);
    } else {
        w = ld<F>(wgt + 2 * idx);
        if (!first && ((idx >> (bit + 1)) & 1u)) { w = fe_mul(w, u); st<F>(wgt + 2 * idx, w); }
    }
    const size_t i = idx & (half - 1);
    const bool hi = (idx >> bit) & 1u;
    Fe<F> v = fe_mul(w, pn(hi ? i : half + i));
    if (canon) v = fe_from_mont(v);
    const Fe<F> zero = fe_zero<F>();
    st<F>(lrsc + 2 * idx, hi ? zero : v);
    st<F>(lrsc + 2 * (stride + idx), hi ? v : zero);
    if (!first && idx < hprev) {  // the folded vectors, once
        st<F>(p_new + 2 * idx, pn(idx));
        st<F>(b_new + 2 * idx, fe_add(ld<F>(b_old + 2 * idx), fe_mul(ld<F>(b_old + 2 * (hprev + idx)), u)));
    }
}
// dst[i] = a[i] + x s[i] (s may be null: a plain copy) and the block sums of dst[i] b[i], one pass: the opening's s(X) -> s' and p + xi s -> p'
// with their values at x3 (consts = (x)).  ipa_fix_constant_kernel then subtracts the value from coefficient 0 -- on the device: the host never
// needs s(x3) or p'(x3) (round 6; each was a synchronisation and two 32-byte copies before).
template <class F>
__global__ void __launch_bounds__(256) ipa_combine_eval_kernel(const uint4* __restrict__ a, const uint4* __restrict__ sv, const uint4* __restrict__ b, uint4* __restrict__ dst, size_t n,
                                                               const IpaConsts consts, uint4* __restrict__ partial) {
    __shared__ Fe<F> sh[256];
    const Fe<F> x = ipa_const<F>(consts, 0);
    Fe<F> acc = fe_zero<F>();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fe<F> v = ld<F>(a + 2 * i);
        if (sv) v = fe_add(v, fe_mul(ld<F>(sv + 2 * i), x));
        st<F>(dst + 2 * i, v);
        acc = fe_add(acc, fe_mul(v, ld<F>(b + 2 * i)));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st<F>(partial + 2 * blockIdx.x, sh[0]);
}
// one workgroup: vec[0] -= sum of the partials; with tail: vec[n] = consts[0] (the blind), vec[n + 1] = 0 (the scalar of u)
template <class F>
__global__ void __launch_bounds__(256) ipa_fix_constant_kernel(const uint4* __restrict__ partial, u32 count, uint4* __restrict__ vec, size_t n, int tail, const IpaConsts consts) {
    __shared__ Fe<F> sh[256];
    Fe<F> acc = fe_zero<F>();
    for (u32 i = threadIdx.x; i < count; i += 256) acc = fe_add(acc, ld<F>(partial + 2 * i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st<F>(vec, fe_sub(ld<F>(vec), sh[0]));
        if (tail) { st<F>(vec + 2 * n, ipa_const<F>(consts, 0)); st<F>(vec + 2 * (n + 1), fe_zero<F>()); }
    }
}

template <class F>
__global__ void __launch_bounds__(256) ipa_round_tails_kernel(const uint4* __restrict__ partial, u32 count, const IpaConsts consts /* rand_l, rand_r, z */,
                                                              uint4* __restrict__ lrsc, size_t n, size_t stride, int canon /* as ipa_round_front_kernel */) {
    __shared__ Fe<F> sh[256];
    const u32 side = blockIdx.x;
    const uint4* p = partial + 2 * (size_t)side * count;
    Fe<F> acc = fe_zero<F>();
    for (u32 i = threadIdx.x; i < count; i += 256) acc = fe_add(acc, ld<F>(p + 2 * i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        Fe<F> r = side ? ipa_const<F>(consts, 1) : ipa_const<F>(consts, 0), vz = fe_mul(sh[0], ipa_const<F>(consts, 2));
        if (canon) { r = fe_from_mont(r); vz = fe_from_mont(vz); }
        st<F>(lrsc + 2 * (side * stride + n), r);
        st<F>(lrsc + 2 * (side * stride + n + 1), vz);
    }
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

// small device -> host read-back through the pinned landing area (callers hold the context lock)
int read_back(void* out, const void* dev, size_t bytes, hipStream_t s) {
    Ctx& c = ctx();
    if (bytes > 4096) {
        TRH_HIP_TRY(hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, s));
        TRH_HIP_TRY(hipStreamSynchronize(s));
        return TRH_OK;
    }
    if (!c.pinned_land) TRH_HIP_TRY(hipHostMalloc(&c.pinned_land, 4096, hipHostMallocDefault));
    TRH_HIP_TRY(hipMemcpyAsync(c.pinned_land, dev, bytes, hipMemcpyDeviceToHost, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));
    memcpy(out, c.pinned_land, bytes);
    return TRH_OK;
}

// small per-call constant staged in the factor ring (see trh_field_scale_rows_dev)
int stage_constant(const void* host, size_t bytes, hipStream_t s, void** dev) {
    Ctx& c = ctx();
    constexpr unsigned SLOTS = 64, SLOT_BYTES = 2048;
    if (bytes > SLOT_BYTES) { set_error("stage_constant: %zu bytes", bytes); return TRH_EINVAL; }
    TRH_TRY(c.factors.ensure(16 * 64 * 32 + SLOTS * SLOT_BYTES));  // the first 32 KiB belong to the scale kernels' ring
    if (!c.pinned_ring) TRH_HIP_TRY(hipHostMalloc(&c.pinned_ring, SLOTS * SLOT_BYTES, hipHostMallocDefault));
    const unsigned k = c.pinned_slot++ % SLOTS;
    if (k == 0 && c.pinned_slot > 1) TRH_HIP_TRY(hipDeviceSynchronize());  // wrap-around: every earlier upload (on whatever stream) has left the pinned slots
    char* hslot = (char*)c.pinned_ring + (size_t)k * SLOT_BYTES;
    char* dslot = (char*)c.factors.p + 16 * 64 * 32 + (size_t)k * SLOT_BYTES;
    memcpy(hslot, host, bytes);
    TRH_HIP_TRY(hipMemcpyAsync(dslot, hslot, bytes, hipMemcpyHostToDevice, s));
    *dev = dslot;
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

// poly::eval_polynomial for `batch` polynomials of n coefficients at one point: the powers x^i are built once
template <class F>
int eval_batch_t(const void* polys, size_t n, size_t batch, const u64* x, hipStream_t s, u64* out) {
    Ctx& c = ctx();
    unsigned blocks = (unsigned)((n + 2047) / 2048);  // eight elements per thread
    if (blocks > 64) blocks = 64;
    if (blocks == 0) blocks = 1;
    TRH_TRY(c.scan.ensure(n * 32 + 32));
    TRH_TRY(c.io.ensure((size_t)batch * (blocks + 1) * 32));
    TRH_TRY((powers_t<F>(c.scan.p, n, x, s)));
    uint4* partial = c.io.as<uint4>();
    uint4* result = partial + 2 * (size_t)batch * blocks;
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {
        const unsigned nb = (unsigned)(batch - b0 < 65535 ? batch - b0 : 65535);
        hipLaunchKernelGGL((eval_batch_kernel<F>), dim3(blocks, nb), dim3(256), 0, s, (const uint4*)polys + b0 * n * 2, c.scan.as<uint4>(), n, partial + 2 * b0 * blocks);
        hipLaunchKernelGGL((sum_partials_batch_kernel<F>), dim3(nb), dim3(256), 0, s, partial + 2 * b0 * blocks, blocks, result + 2 * b0);
    }
    TRH_HIP_TRY(hipGetLastError());
    return read_back(out, result, batch * 32, s);
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

template <class F>
int axpy_t(void* y, const void* x, size_t n, const FeMem& c_mont, hipStream_t s) {
    if (!n) return TRH_OK;
    void* d_c;
    TRH_TRY(stage_constant(&c_mont, 32, s, &d_c));
    hipLaunchKernelGGL((axpy_kernel<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4*)y, (const uint4*)x, n, (const uint4*)d_c);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

// ---------------------------------------------------------------------------------------
// The whole opening: poly::commitment::prover::create_proof with p', b and G' resident on the
// device.  Host-side scalar arithmetic uses the shared field code; the transcript and the prover's
// randomness are callbacks into the caller (BLAKE2b transcript and OsRng on the Rust side).
// ---------------------------------------------------------------------------------------
// the round after which the generators are collapsed (0: never -- no table, a small opening, or option ipa_fold = 0).  Option 1 = the measured
// choice: after six rounds, later where the collapsed set would not fit msm_small_kernel (k = 18: 5 / 6 / 7 / 8 rounds -> 9.3 / 9.0 / 9.2 / 9.6 ms)
uint32_t ipa_fold_level(const MsmFixedBase* fb, uint32_t k) {
    uint32_t r = (uint32_t)opt().ipa_fold;
    if (r == 1) r = k >= 14 ? k - 12 : 0;  // down to 2^12 generators
    return (fb && r > 0 && k >= 14 && ipa_fold_supported(*fb, k, r)) ? r : 0;
}

template <class SF, class BF>
int ipa_create_proof_t(int curve, const trh_bases* gw, const u64* u_xy, uint32_t k, const void* p_poly_dev, const u64* p_blind_m, const u64* x3_m,
                       const void* s_poly_dev, const u64* s_blind_m, const trh_transcript_t* tr, trh_rng_scalar_fn rng, void* rng_ctx,
                       hipStream_t s, u64* out_c, u64* out_f) {
    const size_t n = (size_t)1 << k;
    auto ld = [](const u64* p) { FeMem m; memcpy(&m, p, 32); return fe_load<SF>(m); };
    auto stm = [](const Fe<SF>& v) { FeMem m; fe_store(v, m); return m; };
    const Fe<SF> x3 = ld(x3_m), p_blind = ld(p_blind_m), s_blind = ld(s_blind_m);

    ctx().msm.lean_off_n = 0;  // a shape whose lean sort overflowed keeps the fallback launches for the rest of ITS opening only (msm.hip lean_sort)
    // scratch kept in the context: a hipMalloc / hipFree pair per vector costs more than several rounds
    DevBuf* sc = ctx().ipa;
    DevBuf &b = sc[0], &sp = sc[1], &pp = sc[2], &wgt = sc[3], &lrsc = sc[4], &gwu = sc[5], &gwuz = sc[6], &pp2 = sc[7], &b2 = sc[8];
    // A base set that holds g || w || u (n + 2 points) with fixed-base tables attached (trh_bases_precompute: Params are fixed for
    // the life of a proving key) lets every MSM of the opening run in fixed-base mode: one bucket set instead of one per window,
    // wide windows, no heavy top-window buckets and no Horner over windows on the host between two rounds.
    const bool with_u = gw->n == n + 2;
    const MsmFixedBase* fb = (with_u && gw->d_table) ? &gw->fb : nullptr;
    // A set small enough for msm_small_kernel does not use its table: one launch + a host Horner beats the fixed-base pipeline's chain of ten
    // launches (k = 10: 3.3 -> 2.1 ms per opening, k = 13: 5.4 -> 3.7); level 0 of the table is the set in the accumulation's record format
    const void* small_z = nullptr;
    if (fb && n + 2 <= msm_small_max_pairs() && ctx().window_override == 0) { small_z = fb->table; fb = nullptr; }
    TRH_TRY(b.ensure(n * 32)); TRH_TRY(sp.ensure((n + 2) * 32)); TRH_TRY(pp.ensure(n * 32)); TRH_TRY(wgt.ensure(n * 32)); TRH_TRY(lrsc.ensure(2 * (n + 2) * 32));
    TRH_TRY(pp2.ensure(n * 16 + 32)); TRH_TRY(b2.ensure(n * 16 + 32));  // the folded vectors have at most n / 2 entries
    if (!with_u) { TRH_TRY(gwu.ensure((n + 2) * 64)); TRH_TRY(gwuz.ensure((n + 2) * ZREC)); }
    FeMem x3m = stm(x3);
    TRH_TRY((powers_t<SF>(b.p, n, (const u64*)&x3m, s)));
    // s'(X) = s(X) - s(x3), then its commitment over g ‖ w with the blind appended: one pass copies s and forms its value at x3, one workgroup
    // subtracts it from coefficient 0 and appends the blind (and the zero scalar of u) -- no host turn before the MSM
    FeMem tmp;
    unsigned eblocks = (unsigned)((n + 255) / 256);
    if (eblocks > 512) eblocks = 512;
    TRH_TRY(ctx().io.ensure((size_t)(2 * 512 + 2) * 32));
    {
        IpaConsts k0{};
        hipLaunchKernelGGL((ipa_combine_eval_kernel<SF>), dim3(eblocks), dim3(256), 0, s, (const uint4*)s_poly_dev, (const uint4*)nullptr, (const uint4*)b.p, (uint4*)sp.p, n, k0,
                           ctx().io.as<uint4>());
        FeMem sbm = stm(s_blind);
        IpaConsts kb{};
        memcpy(&kb, &sbm, 32);
        hipLaunchKernelGGL((ipa_fix_constant_kernel<SF>), dim3(1), dim3(256), 0, s, (const uint4*)ctx().io.as<uint4>(), eblocks, (uint4*)sp.p, n, 1, kb);
        TRH_HIP_TRY(hipGetLastError());
    }
    u64 pt[12];
    if (fb) {  // the scalar of u is zero: the full-range (table) path
        ctx().msm.dense_hint = true;  // s(X) is uniformly random: no sparse classification
        const int rc_s = msm_enqueue(curve, gw->d_xy, gw->d_z, sp.p, n + 2, 1, n + 2, 1, s, fb);
        ctx().msm.dense_hint = false;
        TRH_TRY(rc_s);
    } else if (small_z)
    TRH_TRY(msm_enqueue(curve, gw->d_xy, small_z, sp.p, n + 2, 1, n + 2, 1, s));
    else
    TRH_TRY(msm_enqueue(curve, gw->d_xy, gw->d_z, sp.p, n + 1, 1, n + 1, 1, s));
    TRH_TRY(msm_finish(curve, s, pt, 1));
    tr->write_point(tr->ctx, pt);
    tr->squeeze_challenge_scalar(tr->ctx, (u64*)&tmp);
    const Fe<SF> xi = fe_load<SF>(tmp);
    tr->squeeze_challenge_scalar(tr->ctx, (u64*)&tmp);
    const Fe<SF> z = fe_load<SF>(tmp);
    // p'(X) = p(X) + xi s'(X) - v, v its value at x3: the same two launches
    {
        FeMem xim = stm(xi);
        IpaConsts kx{};
        memcpy(&kx, &xim, 32);
        hipLaunchKernelGGL((ipa_combine_eval_kernel<SF>), dim3(eblocks), dim3(256), 0, s, (const uint4*)p_poly_dev, (const uint4*)sp.p, (const uint4*)b.p, (uint4*)pp.p, n, kx,
                           ctx().io.as<uint4>());
        IpaConsts k0{};
        hipLaunchKernelGGL((ipa_fix_constant_kernel<SF>), dim3(1), dim3(256), 0, s, (const uint4*)ctx().io.as<uint4>(), eblocks, (uint4*)pp.p, n, 0, k0);
        TRH_HIP_TRY(hipGetLastError());
    }
    Fe<SF> f = fe_add(fe_mul(s_blind, xi), p_blind);
    // weights of the original generators inside the (virtual) folded ones; the bases of the round MSMs are
    // g (n points) followed by w and u, so that [rand] W + [value z] U ride in the same MSM as the main sum
    // (all ones at first: the front launch of round 0 takes them as such and round 1's writes them -- no fill)
    const void* round_xy = gw->d_xy;
    const void* round_z = nullptr;
    if (!with_u) {
        TRH_HIP_TRY(hipMemcpyAsync(gwu.p, gw->d_xy, (n + 1) * 64, hipMemcpyDeviceToDevice, s));
        TRH_HIP_TRY(hipMemcpy((char*)gwu.p + (n + 1) * 64, u_xy, 64, hipMemcpyHostToDevice));
        TRH_TRY(msm_convert_bases(curve, gwu.p, gwuz.p, n + 2, s));
        round_xy = gwu.p; round_z = gwuz.p;
    } else if (!fb) {
        round_z = small_z ? small_z : gw->d_z;  // the handle's converted copy when it has one; otherwise the MSM converts per call
    }
    size_t stride = n + 2;
    // Hybrid (round 6, ipafold.hip): the first r rounds run over the 2^k original bases with the folds as weights (a full-size fixed-base MSM each);
    // then the generators are collapsed for real -- r folds at once from the table -- and the remaining k - r rounds are MSMs over m + 2 = 2^(k - r) + 2
    // points, a fraction of a full-size round each.  Only worth it where a full-size round costs much more than a small one.
    const uint32_t fold_at = ipa_fold_level(fb, k);
    const MsmFixedBase* round_fb = fb;
    size_t ncur = n;  // generators of the round MSMs
    u64 u_hist[16 * 4];  // the challenges the collapse needs

    // the round MSMs are batches of two with half of the scalars zero: their time is the latency of the sort / reduction chain,
    // which narrow windows shorten (measured: k = 18 -> c = 10 gives 24.4 ms against 26.7 at the table's 15; k = 14 -> 8)
    struct WindowGuard {
        int& slot; int saved;
        WindowGuard(int& s_, int v) : slot(s_), saved(s_) { if (!saved) slot = v; }
        ~WindowGuard() { slot = saved; }
    } window_guard(ctx().window_override, (!fb && k >= 16 && k <= 18) ? (int)k - 8 : 0);  // k = 20: the table's 15 is better again (60 vs 68 ms)

    // per round three launches in front of the MSM (the previous round's folds applied on the fly + the scalar rows; both inner products; the tail
    // scalars -- the round's constants travel as kernel arguments), the MSM, the transcript.  No host synchronisation before the MSM: the host
    // never needs value_l / value_r.  (Round 5 had four launches here, the first version 14 small operations.)
    FeMem zm = stm(z);
    // TRH_TRACE bit 1: where the host's part of a round goes (averages over the rounds, microseconds, to stderr)
    const bool ipa_trace = (opt().trace & 2) != 0;
    double tr_acc[6] = {0, 0, 0, 0, 0, 0};
    auto tnow = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    unsigned fblocks = (unsigned)((n + 255) / 256);
    uint4* const partial = ctx().io.as<uint4>();
    // p' and b live in ping-pong pairs from round 1 on (a round reads the vectors the previous one left and writes their folds to the other buffer)
    void* p_cur = pp.p; void* b_cur = b.p;
    void* p_nxt = pp2.p; void* b_nxt = b2.p;
    Fe<SF> u_prev = fe_one<SF>(), u_prev_inv = fe_one<SF>();
    for (uint32_t j = 0; j < k; ++j) {
        const double t_round0 = ipa_trace ? tnow() : 0;
        const size_t half = (size_t)1 << (k - j - 1);
        const u32 bit = k - j - 1;
        Fe<SF> rnd[2];
        rng(rng_ctx, (u64*)&tmp); rnd[0] = fe_load<SF>(tmp);
        rng(rng_ctx, (u64*)&tmp); rnd[1] = fe_load<SF>(tmp);
        if (fold_at && j == fold_at) {  // G'' || w || u from the table, after the challenges of rounds 0 .. r - 1; from here on the rounds are MSMs over m + 2 points
            // (inline: on a stream of its own under the next full-size rounds the collapse gained nothing -- profiles/r06_ipa_fold_overlap_ab.txt)
            const size_t m = (size_t)1 << (k - fold_at);
            TRH_TRY(gwu.ensure((m + 2) * 64)); TRH_TRY(gwuz.ensure((m + 2) * ZREC));
            TRH_TRY(ipa_fold_generators(curve, *fb, gw->n, k, fold_at, u_hist, gwu.p, gwuz.p, s));
            TRH_HIP_TRY(hipMemcpyAsync((char*)gwu.p + m * 64, (const char*)gw->d_xy + n * 64, 2 * 64, hipMemcpyDeviceToDevice, s));
            TRH_HIP_TRY(hipMemcpyAsync((char*)gwuz.p + m * ZREC, (const char*)fb->table + n * ZREC, 2 * ZREC, hipMemcpyDeviceToDevice, s));  // level 0 of the table = the bases
            ncur = m; stride = m + 2; fblocks = (unsigned)((m + 255) / 256);
            round_xy = gwu.p; round_z = gwuz.p; round_fb = nullptr;
            ++ctx().ipa_collapses;
        }
        const int canon = (!round_fb && ncur + 2 <= msm_small_max_pairs() && ctx().window_override == 0) ? 1 : 0;  // this round's MSM is msm_small_kernel's
        {
            FeMem cst[2] = {stm(u_prev_inv), stm(u_prev)};
            IpaConsts kc{};
            memcpy(&kc, cst, sizeof(cst));
            hipLaunchKernelGGL((ipa_round_front_kernel<SF>), dim3(fblocks), dim3(256), 0, s, (const uint4*)p_cur, (const uint4*)b_cur, (uint4*)p_nxt, (uint4*)b_nxt, (uint4*)wgt.p, (uint4*)lrsc.p,
                               ncur, half, bit, stride, j == 0 ? 1 : 0, (j <= 1 || (fold_at && j - fold_at <= 1)) ? 1 : 0, canon, kc);
            if (j > 0) { void* t = p_cur; p_cur = p_nxt; p_nxt = t; t = b_cur; b_cur = b_nxt; b_nxt = t; }
            // value_l = <p'[half ..], b[.. half]>, value_r = <p'[.. half], b[half ..]> over the folded vectors, then the tail scalars [rand] W, [value z] U
            unsigned blocks = (unsigned)((half + 255) / 256);
            if (blocks > 512) blocks = 512;
            hipLaunchKernelGGL((inner_product2_kernel<SF>), dim3(blocks, 2), dim3(256), 0, s, (const uint4*)((const char*)p_cur + half * 32), (const uint4*)b_cur, (const uint4*)p_cur,
                               (const uint4*)((const char*)b_cur + half * 32), half, partial);
            FeMem cst3[3] = {stm(rnd[0]), stm(rnd[1]), zm};
            IpaConsts kt{};
            memcpy(&kt, cst3, sizeof(cst3));
            hipLaunchKernelGGL((ipa_round_tails_kernel<SF>), dim3(2), dim3(256), 0, s, partial, blocks, kt, (uint4*)lrsc.p, ncur, stride, canon);
            TRH_HIP_TRY(hipGetLastError());
        }
        u64 lrb[24], lr[2][12];
        ctx().msm.dense_hint = true;  // p' . w: full-size values on half the rows
        const int rc_r = msm_enqueue(curve, round_xy, round_z, lrsc.p, ncur + 2, 2, stride, canon ? 0 : 1, s, round_fb);
        ctx().msm.dense_hint = false;
        TRH_TRY(rc_r);
        const double t_enq = ipa_trace ? tnow() : 0;
        TRH_TRY(msm_finish(curve, s, lrb, 2));
        const double t_fin = ipa_trace ? tnow() : 0;
        memcpy(lr[0], lrb, 96); memcpy(lr[1], lrb + 12, 96);
        tr->write_point(tr->ctx, lr[0]);
        tr->write_point(tr->ctx, lr[1]);
        tr->squeeze_challenge_scalar(tr->ctx, (u64*)&tmp);
        const double t_tr = ipa_trace ? tnow() : 0;
        const Fe<SF> u_j = fe_load<SF>(tmp);
        if (fe_is_zero(u_j)) { set_error("ipa_create_proof: round %u challenge is zero (the Rust prover's u_j.invert().unwrap() panics here)", j); return TRH_EINVAL; }
        // u_j^-1 on the host's 4 x 64-bit Montgomery code (hostcombine.h): the nine-limb form the device code uses costs the host three times as much
        hostcombine::H uh;
        memcpy(&uh, &tmp, 32);
        const hostcombine::H uih = hostcombine::inv<SF>(uh);
        FeMem uim;
        memcpy(&uim, &uih, 32);
        const Fe<SF> u_inv = fe_load<SF>(uim);
        if (j < 16) memcpy(u_hist + 4 * j, &tmp, 32);
        u_prev = u_j; u_prev_inv = u_inv;  // folded into the next round's front launch (or by the launch behind the loop)
        f = fe_add(f, fe_add(fe_mul(rnd[0], u_inv), fe_mul(rnd[1], u_j)));
        if (ipa_trace) {
            const double t_end = tnow();
            tr_acc[0] += t_enq - t_round0; tr_acc[1] += t_fin - t_enq; tr_acc[2] += t_tr - t_fin; tr_acc[3] += t_end - t_tr; tr_acc[4] += t_end - t_round0;
        }
    }
    if (k > 0) {  // the last round's fold: p'[0] += u^-1 p'[1] (b and the weights are not read again, the kernel folds them too)
        FeMem cst[2] = {stm(u_prev_inv), stm(u_prev)};
        IpaConsts kc{};
        memcpy(&kc, cst, sizeof(cst));
        hipLaunchKernelGGL((ipa_round_update_kernel<SF>), dim3((unsigned)((ncur + 255) / 256)), dim3(256), 0, s, (uint4*)p_cur, (uint4*)b_cur, (uint4*)wgt.p, ncur, (size_t)1, 0u, kc);
        TRH_HIP_TRY(hipGetLastError());
    }
    if (ipa_trace)
        fprintf(stderr, "[trh ipa] k = %u, per round (us): enqueue %.1f, wait for the MSM (sync + host combine) %.1f, transcript callbacks %.1f, inversion + update launch %.1f, round %.1f\n",
                k, tr_acc[0] / k, tr_acc[1] / k, tr_acc[2] / k, tr_acc[3] / k, tr_acc[4] / k);
    TRH_HIP_TRY(hipStreamSynchronize(s));
    TRH_HIP_TRY(hipMemcpy(&tmp, p_cur, 32, hipMemcpyDeviceToHost));
    FeMem fm = stm(f);
    tr->write_scalar(tr->ctx, (const u64*)&tmp);
    tr->write_scalar(tr->ctx, (const u64*)&fm);
    if (out_c) memcpy(out_c, &tmp, 32);
    if (out_f) memcpy(out_f, &fm, 32);
    return TRH_OK;
}

}  // namespace

// what an opening over this base set allocates, at setup time (trh_bases_reserve): the first proof of a process then finds its vectors, the
// collapse's buckets and the small MSM's landing area in place
int ipa_reserve(int curve, const trh_bases* gw, uint32_t k) {
    const size_t n = (size_t)1 << k;
    Ctx& c = ctx();
    DevBuf* sc = c.ipa;
    TRH_TRY(sc[0].ensure(n * 32)); TRH_TRY(sc[1].ensure((n + 2) * 32)); TRH_TRY(sc[2].ensure(n * 32)); TRH_TRY(sc[3].ensure(n * 32)); TRH_TRY(sc[4].ensure(2 * (n + 2) * 32));
    TRH_TRY(sc[7].ensure(n * 16 + 32)); TRH_TRY(sc[8].ensure(n * 16 + 32));
    TRH_TRY(c.io.ensure((size_t)(2 * 512 + 2) * 32));
    const MsmFixedBase* fb = gw->d_table ? &gw->fb : nullptr;
    const uint32_t fold_at = ipa_fold_level(fb, k);
    if (!fold_at) return TRH_OK;
    const size_t m = (size_t)1 << (k - fold_at);
    TRH_TRY(sc[5].ensure((m + 2) * 64)); TRH_TRY(sc[6].ensure((m + 2) * ZREC));
    TRH_TRY(ipa_fold_reserve(*fb, k, fold_at));
    if (!c.helper && !c.helper_failed) {  // msm_finish's second Horner thread
        try { c.helper = new HostHelper(); } catch (...) { c.helper_failed = true; }
    }
    c.msm.reserve_only = true;
    const int rc = msm_enqueue(curve, sc[5].p, sc[6].p, sc[4].p, m + 2, 2, m + 2, 1, nullptr);
    c.msm.reserve_only = false;
    return rc;
}
}  // namespace trh

using namespace trh;

extern "C" {

int trh_poly_eval_batch_dev(int field, const void* polys_dev, size_t n, size_t batch, const uint64_t point[4], void* stream, uint64_t* out) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!out || !point || (n && batch && !polys_dev)) { set_error("poly_eval: null pointer"); return TRH_EINVAL; }
    if (!batch) return TRH_OK;
    if (!n) { memset(out, 0, batch * 32); return TRH_OK; }
    TRH_ENTER(stream);
    Range range("trh_poly_eval_batch_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return eval_batch_t<FpParams>(polys_dev, n, batch, point, (hipStream_t)stream, out);
    return eval_batch_t<FqParams>(polys_dev, n, batch, point, (hipStream_t)stream, out);
}

int trh_field_inner_product_dev(int field, const void* a_dev, const void* b_dev, size_t n, void* stream, uint64_t out[4]) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!out || (n && (!a_dev || !b_dev))) { set_error("inner_product: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_field_inner_product_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return inner_product_t<FpParams>(a_dev, b_dev, n, (hipStream_t)stream, out);
    return inner_product_t<FqParams>(a_dev, b_dev, n, (hipStream_t)stream, out);
}

int trh_field_axpy_dev(int field, void* y_dev, const void* x_dev, size_t n, const uint64_t c_mont[4], void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!c_mont || (n && (!y_dev || !x_dev))) { set_error("axpy: null pointer"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_field_axpy_dev");
    Ctx& c = ctx();
    (void)c;
    FeMem cm;
    memcpy(&cm, c_mont, 32);
    if (field == TRH_FP) return axpy_t<FpParams>(y_dev, x_dev, n, cm, (hipStream_t)stream);
    return axpy_t<FqParams>(y_dev, x_dev, n, cm, (hipStream_t)stream);
}

int trh_field_powers_dev(int field, void* out_dev, size_t n, const uint64_t x_mont[4], void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!x_mont || (n && !out_dev)) { set_error("powers: null pointer"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_field_powers_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return powers_t<FpParams>(out_dev, n, x_mont, (hipStream_t)stream);
    return powers_t<FqParams>(out_dev, n, x_mont, (hipStream_t)stream);
}

int trh_bases_fold_dev(int curve, void* g_lo_dev, const void* g_hi_dev, size_t half, const uint64_t u_mont[4], void* stream) {
    TRH_TRY(require_init());
    if (curve != TRH_PALLAS && curve != TRH_VESTA) { set_error("unknown curve id %d", curve); return TRH_EINVAL; }
    if (!u_mont || (half && (!g_lo_dev || !g_hi_dev))) { set_error("bases_fold: null pointer"); return TRH_EINVAL; }
    if (!half) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_bases_fold_dev");
    Ctx& c = ctx();
    (void)c;
    // pallas: scalar field Fq, base field Fp
    if (curve == TRH_PALLAS) return bases_fold_t<FqParams, FpParams>(g_lo_dev, g_hi_dev, half, u_mont, (hipStream_t)stream);
    return bases_fold_t<FpParams, FqParams>(g_lo_dev, g_hi_dev, half, u_mont, (hipStream_t)stream);
}

int trh_ipa_create_proof(trh_bases_t g_w, const uint64_t u_xy[8], uint32_t k, const void* p_poly_dev, const uint64_t p_blind[4], const uint64_t x3[4],
                         const void* s_poly_dev, const uint64_t s_blind[4], const trh_transcript_t* transcript, trh_rng_scalar_fn rng, void* rng_ctx,
                         void* stream, uint64_t out_c[4], uint64_t out_f[4]) {
    TRH_TRY(require_init());
    if (!g_w || !u_xy || !p_poly_dev || !p_blind || !x3 || !s_poly_dev || !s_blind || !transcript || !rng ||
        !transcript->write_point || !transcript->write_scalar || !transcript->squeeze_challenge_scalar) { set_error("ipa_create_proof: null pointer"); return TRH_EINVAL; }
    if (k > 26 || (g_w->n != ((size_t)1 << k) + 1 && g_w->n != ((size_t)1 << k) + 2)) { set_error("ipa_create_proof: bases must hold g (2^k points) followed by w (and optionally u)"); return TRH_EINVAL; }
    if (!g_w->shards.empty()) { set_error("ipa_create_proof: needs a base set on one device (the rounds are sequential: SURVEY 8e)"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_ipa_create_proof");
    if (g_w->owner && g_w->owner->device != ctx().device) { set_error("ipa_create_proof: the base set lives on another device than the calling context"); return TRH_EINVAL; }
    if (ctx().msm.pending_curve >= 0) { set_error("ipa_create_proof: this context has an enqueued MSM that was not finished"); return TRH_EBUSY; }
    if (g_w->n == ((size_t)1 << k) + 2) {  // g || w || u: the resident u must be the caller's
        uint64_t last[8];
        TRH_HIP_TRY(hipMemcpy(last, (const char*)g_w->d_xy + (g_w->n - 1) * 64, 64, hipMemcpyDeviceToHost));
        if (memcmp(last, u_xy, 64) != 0) { set_error("ipa_create_proof: the last point of a g || w || u base set differs from u"); return TRH_EINVAL; }
    }
    if (g_w->curve == TRH_PALLAS)
        return ipa_create_proof_t<FqParams, FpParams>(TRH_PALLAS, g_w, u_xy, k, p_poly_dev, p_blind, x3, s_poly_dev, s_blind, transcript, rng, rng_ctx, (hipStream_t)stream, out_c, out_f);
    return ipa_create_proof_t<FpParams, FqParams>(TRH_VESTA, g_w, u_xy, k, p_poly_dev, p_blind, x3, s_poly_dev, s_blind, transcript, rng, rng_ctx, (hipStream_t)stream, out_c, out_f);
}

}  // extern "C"
