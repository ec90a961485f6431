// Grand-product building blocks of the permutation / lookup arguments (halo2_proofs 0.2.0
// plonk/permutation/prover.rs and plonk/lookup/prover.rs, reached from create_proof --
// /root/reference/src/test_utils.rs:41-49): the product columns are
//     z[0] = 1,   z[i] = prod_{j < i} numerator[j] / denominator[j],
// computed in Rust as `batch_invert` of the denominators followed by a running product.  These are
// the two device primitives: an in-place batch inversion (Montgomery's trick per thread chunk, zeros
// left as zeros exactly like ff::BatchInvert) and an exclusive prefix product.
#include <string.h>

#include "ctx.h"

namespace trh {
namespace {

template <class F>
__device__ __forceinline__ Fe<F> ldf(const uint4* p) {
    uint4 a = p[0], b = p[1];
    return fe_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
template <class F>
__device__ __forceinline__ void stf(uint4* p, const Fe<F>& v) {
    u32 w[8];
    fe_store(v, w);
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

constexpr int INV_CHUNK = 64;  // elements per thread: one field inversion (~380 multiplies) per chunk

// a[i] <- a[i]^-1 (0 stays 0), or num[i] * a[i]^-1 with a numerator column; scratch holds the running products of the chunk.
// Inverses are element-wise facts, so a thread's chunk need not be contiguous: thread t of a workgroup takes the elements
// base + i * 256 + t (i < INV_CHUNK) and every load and store of a wave is one contiguous 2 KiB run
template <class F>
__global__ void __launch_bounds__(256) batch_invert_kernel(uint4* __restrict__ a, uint4* __restrict__ scratch, const uint4* __restrict__ num, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * INV_CHUNK) + threadIdx.x;
    if (base >= n) return;
    const size_t left = (n - base + 255) / 256;
    const int cnt = left < (size_t)INV_CHUNK ? (int)left : INV_CHUNK;
    Fe<F> acc = fe_one<F>();
    for (int i = 0; i < cnt; ++i) {
        const size_t e = base + (size_t)i * 256;
        stf<F>(scratch + 2 * e, acc);  // product of the non-zero elements before this one
        const Fe<F> v = ldf<F>(a + 2 * e);
        if (!fe_is_zero(v)) acc = fe_mul(acc, v);
    }
    Fe<F> inv = fe_inv(acc);
    for (int i = cnt; i-- > 0;) {
        const size_t e = base + (size_t)i * 256;
        const Fe<F> v = ldf<F>(a + 2 * e);
        if (fe_is_zero(v)) continue;
        Fe<F> r = fe_mul(inv, ldf<F>(scratch + 2 * e));
        if (num) r = fe_mul(r, ldf<F>(num + 2 * e));
        stf<F>(a + 2 * e, r);
        inv = fe_mul(inv, v);
    }
}

// out[r][i] = prod over the terms t of row r of (x_t[i] + c_t * y_t[i] + g_t)   (y_t null: x_t[i] + g_t) -- the numerator and
// denominator products of the permutation argument's chunks and of the lookup argument, every row of a proof in one launch
struct DevTerm {
    const uint4* x;
    const uint4* y;
    FeMem c, g;
};
template <class F>
__global__ void __launch_bounds__(256) product_terms_kernel(const DevTerm* __restrict__ terms, const u32* __restrict__ row_start, size_t n, uint4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 r = blockIdx.y;
    const u32 t0 = row_start[r], t1 = row_start[r + 1];
    Fe<F> acc = fe_one<F>();
    for (u32 t = t0; t < t1; ++t) {  // uniform: the descriptors are scalar loads
        const DevTerm& d = terms[t];
        Fe<F> v = fe_add(ldf<F>(d.x + 2 * i), fe_load<F>(d.g));
        if (d.y) v = fe_add(v, fe_mul(ldf<F>(d.y + 2 * i), fe_load<F>(d.c)));
        acc = t == t0 ? v : fe_mul(acc, v);
    }
    stf<F>(out + 2 * ((size_t)r * n + i), acc);
}

constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_BLOCK = 256 * SCAN_PER_THREAD;

// exclusive scan of 256 values in LDS (Hillis-Steele); returns the exclusive prefix of this thread and the block total
// the scanned operation: running product (grand products) or running sum (kate_division, see multiopen.py)
template <class F> struct OpMul {
    static __device__ __forceinline__ Fe<F> id() { return fe_one<F>(); }
    static __device__ __forceinline__ Fe<F> op(const Fe<F>& a, const Fe<F>& b) { return fe_mul(a, b); }
};
template <class F> struct OpAdd {
    static __device__ __forceinline__ Fe<F> id() { return fe_zero<F>(); }
    static __device__ __forceinline__ Fe<F> op(const Fe<F>& a, const Fe<F>& b) { return fe_add(a, b); }
};

template <class F, class OP>
__device__ __forceinline__ Fe<F> block_exclusive_scan(Fe<F>* sh, const Fe<F>& mine, Fe<F>& total) {
    const int t = threadIdx.x;
    sh[t] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        Fe<F> v = OP::id();
        const bool take = t >= off;
        if (take) v = sh[t - off];
        __syncthreads();
        if (take) sh[t] = OP::op(sh[t], v);
        __syncthreads();
    }
    total = sh[255];
    const Fe<F> incl_prev = t ? sh[t - 1] : OP::id();
    __syncthreads();
    return incl_prev;
}

// phase 1: product of each block of SCAN_BLOCK elements
template <class F, class OP>
__global__ void __launch_bounds__(256) scan_block_totals_kernel(const uint4* __restrict__ a, size_t n, uint4* __restrict__ totals) {
    __shared__ Fe<F> sh[256];
    a += (size_t)blockIdx.y * n * 2; totals += (size_t)blockIdx.y * gridDim.x * 2;  // independent rows of n elements
    const size_t lo = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    Fe<F> p = OP::id();
    for (int k = 0; k < SCAN_PER_THREAD; ++k)
        if (lo + k < n) p = OP::op(p, ldf<F>(a + 2 * (lo + k)));
    Fe<F> total;
    block_exclusive_scan<F, OP>(sh, p, total);
    if (threadIdx.x == 0) stf<F>(totals + 2 * blockIdx.x, total);
}
// phase 2 (one workgroup): exclusive scan of the block totals, in place
template <class F, class OP>
__global__ void __launch_bounds__(256) scan_totals_kernel(uint4* __restrict__ totals, u32 count) {
    __shared__ Fe<F> sh[256];
    totals += (size_t)blockIdx.x * count * 2;  // one workgroup per row
    const u32 per = (count + 255u) / 256u;
    const u32 lo = threadIdx.x * per;
    Fe<F> p = OP::id();
    for (u32 k = 0; k < per; ++k)
        if (lo + k < count) p = OP::op(p, ldf<F>(totals + 2 * (lo + k)));
    Fe<F> total;
    Fe<F> run = block_exclusive_scan<F, OP>(sh, p, total);
    for (u32 k = 0; k < per; ++k) {
        if (lo + k >= count) break;
        const Fe<F> v = ldf<F>(totals + 2 * (lo + k));
        stf<F>(totals + 2 * (lo + k), run);
        run = OP::op(run, v);
    }
}
// phase 3: out[i] = prod_{j < i} a[j]
template <class F, class OP>
__global__ void __launch_bounds__(256) scan_apply_kernel(const uint4* __restrict__ a, uint4* __restrict__ out, size_t n, const uint4* __restrict__ totals) {
    __shared__ Fe<F> sh[256];
    a += (size_t)blockIdx.y * n * 2; out += (size_t)blockIdx.y * n * 2; totals += (size_t)blockIdx.y * gridDim.x * 2;
    const size_t lo = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    Fe<F> vals[SCAN_PER_THREAD];
    Fe<F> p = OP::id();
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        vals[k] = lo + k < n ? ldf<F>(a + 2 * (lo + k)) : OP::id();
        p = OP::op(p, vals[k]);
    }
    Fe<F> total;
    Fe<F> run = OP::op(block_exclusive_scan<F, OP>(sh, p, total), ldf<F>(totals + 2 * blockIdx.x));
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        if (lo + k < n) stf<F>(out + 2 * (lo + k), run);
        run = OP::op(run, vals[k]);
    }
}

template <class F, class OP>
int prefix_scan_t(const void* a, void* out, size_t n, hipStream_t s, size_t rows = 1) {
    Ctx& c = ctx();
    const unsigned blocks = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    for (size_t r0 = 0; r0 < rows; r0 += 32768) {  // grid.y limit
        const unsigned nr = (unsigned)(rows - r0 < 32768 ? rows - r0 : 32768);
        TRH_TRY(c.scan.ensure((size_t)nr * blocks * 32 + 32));
        const uint4* ar = (const uint4*)a + r0 * n * 2;
        uint4* outr = (uint4*)out + r0 * n * 2;
        hipLaunchKernelGGL((scan_block_totals_kernel<F, OP>), dim3(blocks, nr), dim3(256), 0, s, ar, n, c.scan.as<uint4>());
        hipLaunchKernelGGL((scan_totals_kernel<F, OP>), dim3(nr), dim3(256), 0, s, c.scan.as<uint4>(), blocks);
        hipLaunchKernelGGL((scan_apply_kernel<F, OP>), dim3(blocks, nr), dim3(256), 0, s, ar, outr, n, c.scan.as<uint4>());
    }
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}
// ---- multiopen building blocks (halo2_proofs 0.2.0 poly/multiopen/prover.rs) -----------------------------
// out[i] = sum_b coeff[b] * polys[b][i]: the x1 / x4 linear combinations of the queried polynomials
template <class F>
__global__ void __launch_bounds__(256) lincomb_kernel(const uint4* __restrict__ polys, size_t n, u32 batch, const uint4* __restrict__ coeffs, uint4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> acc = fe_zero<F>();
    for (u32 b = 0; b < batch; ++b) acc = fe_add(acc, fe_mul(ldf<F>(polys + 2 * ((size_t)b * n + i)), ldf<F>(coeffs + 2 * b)));
    stf<F>(out + 2 * i, acc);
}
// kate_division, step 1: t[m] = a[m] * z^m (pz = the powers of z)
template <class F>
__global__ void __launch_bounds__(256) mul_pointwise_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stf<F>(out + 2 * i, fe_mul(ldf<F>(a + 2 * i), ldf<F>(b + 2 * i)));
}
// step 3: q[i - 1] = (total - P[i]) * zinv^i for i = 1..n-1, P = exclusive prefix sums of t, total = P[n-1] + t[n-1]
template <class F>
__global__ void __launch_bounds__(256) kate_finish_kernel(const uint4* __restrict__ t, const uint4* __restrict__ P, const uint4* __restrict__ pzinv, uint4* __restrict__ q, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i >= n) return;
    const Fe<F> total = fe_add(ldf<F>(P + 2 * (n - 1)), ldf<F>(t + 2 * (n - 1)));
    stf<F>(q + 2 * (i - 1), fe_mul(fe_sub(total, ldf<F>(P + 2 * i)), ldf<F>(pzinv + 2 * i)));
}

template <class F>
int batch_invert_t(void* a, const void* num, size_t n, hipStream_t s) {
    Ctx& c = ctx();
    TRH_TRY(c.scan2.ensure(n * 32 + 32));
    const size_t per_block = (size_t)256 * INV_CHUNK;
    hipLaunchKernelGGL((batch_invert_kernel<F>), dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), 0, s, (uint4*)a, c.scan2.as<uint4>(), (const uint4*)num, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

}  // namespace
}  // namespace trh

using namespace trh;

extern "C" {

int trh_field_batch_invert_dev(int field, void* a_dev, size_t n, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (n && !a_dev) { set_error("batch_invert: null pointer"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_field_batch_invert_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return batch_invert_t<FpParams>(a_dev, nullptr, n, (hipStream_t)stream);
    return batch_invert_t<FqParams>(a_dev, nullptr, n, (hipStream_t)stream);
}

int trh_field_batch_invert_mul_dev(int field, void* a_dev, const void* num_dev, size_t n, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (n && (!a_dev || !num_dev)) { set_error("batch_invert_mul: null pointer"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_field_batch_invert_mul_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return batch_invert_t<FpParams>(a_dev, num_dev, n, (hipStream_t)stream);
    return batch_invert_t<FqParams>(a_dev, num_dev, n, (hipStream_t)stream);
}

int trh_product_terms_dev(int field, const trh_product_term_t* terms, const uint32_t* row_start, uint32_t rows, size_t n, void* out_dev, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!rows || !n) return TRH_OK;
    if (!terms || !row_start || !out_dev) { set_error("product_terms: null pointer"); return TRH_EINVAL; }
    if (rows > 32768) { set_error("product_terms: more than 32768 rows"); return TRH_EINVAL; }
    const uint32_t n_terms = row_start[rows];
    if (row_start[0] != 0 || n_terms > (1u << 20)) { set_error("product_terms: bad row_start"); return TRH_EINVAL; }
    for (uint32_t r = 0; r < rows; ++r)
        if (row_start[r + 1] <= row_start[r]) { set_error("product_terms: row %u has no terms", r); return TRH_EINVAL; }
    for (uint32_t t = 0; t < n_terms; ++t)
        if (!terms[t].x) { set_error("product_terms: term %u has a null column", t); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_product_terms_dev");
    Ctx& c = ctx();
    hipStream_t s = (hipStream_t)stream;
    static_assert(sizeof(DevTerm) == sizeof(trh_product_term_t), "descriptor layout");
    const size_t term_bytes = (size_t)n_terms * sizeof(DevTerm), start_bytes = ((size_t)rows + 1) * 4;
    TRH_TRY(c.scan.ensure(term_bytes + start_bytes));
    TRH_HIP_TRY(hipMemcpyAsync(c.scan.p, terms, term_bytes, hipMemcpyHostToDevice, s));
    TRH_HIP_TRY(hipMemcpyAsync((char*)c.scan.p + term_bytes, row_start, start_bytes, hipMemcpyHostToDevice, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));  // the caller's descriptor arrays may be reused
    const dim3 grid((unsigned)((n + 255) / 256), rows);
    if (field == TRH_FP) hipLaunchKernelGGL((product_terms_kernel<FpParams>), grid, dim3(256), 0, s, (const DevTerm*)c.scan.p, (const u32*)((char*)c.scan.p + term_bytes), n, (uint4*)out_dev);
    else hipLaunchKernelGGL((product_terms_kernel<FqParams>), grid, dim3(256), 0, s, (const DevTerm*)c.scan.p, (const u32*)((char*)c.scan.p + term_bytes), n, (uint4*)out_dev);
    TRH_HIP_TRY(hipGetLastError());
    TRH_HIP_TRY(hipStreamSynchronize(s));  // c.scan is shared scratch
    return TRH_OK;
}

int trh_field_prefix_product_dev(int field, const void* a_dev, void* out_dev, size_t n, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (n && (!a_dev || !out_dev)) { set_error("prefix_product: null pointer"); return TRH_EINVAL; }
    if (a_dev == out_dev) { set_error("prefix_product: in-place operation is not supported"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_field_prefix_product_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return prefix_scan_t<FpParams, OpMul<FpParams>>(a_dev, out_dev, n, (hipStream_t)stream);
    return prefix_scan_t<FqParams, OpMul<FqParams>>(a_dev, out_dev, n, (hipStream_t)stream);
}

int trh_poly_lincomb_dev(int field, const void* polys_dev, size_t n, size_t batch, const uint64_t* coeffs_host, void* out_dev, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (n && (!out_dev || (batch && (!polys_dev || !coeffs_host)))) { set_error("poly_lincomb: null pointer"); return TRH_EINVAL; }
    if (batch > ((size_t)1 << 20)) { set_error("poly_lincomb: batch too large"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_poly_lincomb_dev");
    Ctx& c = ctx();
    (void)c;
    hipStream_t s = (hipStream_t)stream;
    TRH_TRY(c.scan.ensure((batch ? batch : 1) * 32));
    if (batch) TRH_HIP_TRY(hipMemcpyAsync(c.scan.p, coeffs_host, batch * 32, hipMemcpyHostToDevice, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));  // the caller's coefficient buffer may be reused
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (field == TRH_FP) hipLaunchKernelGGL((lincomb_kernel<FpParams>), dim3(gb), dim3(256), 0, s, (const uint4*)polys_dev, n, (u32)batch, c.scan.as<uint4>(), (uint4*)out_dev);
    else hipLaunchKernelGGL((lincomb_kernel<FqParams>), dim3(gb), dim3(256), 0, s, (const uint4*)polys_dev, n, (u32)batch, c.scan.as<uint4>(), (uint4*)out_dev);
    TRH_HIP_TRY(hipGetLastError());
    TRH_HIP_TRY(hipStreamSynchronize(s));  // c.scan is shared scratch
    return TRH_OK;
}

/* poly::kate_division(a, z): the quotient of a(X) by (X - z), remainder dropped; pz / pzinv: the powers of z and of z^-1
 * (n elements each, device; trh_field_powers_dev), scratch: 2 n elements of device memory */
int trh_poly_kate_division_dev(int field, const void* a_dev, size_t n, const void* pz_dev, const void* pzinv_dev, void* scratch_dev, void* q_dev, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (n < 2) return TRH_OK;  // a constant has an empty quotient
    if (!a_dev || !pz_dev || !pzinv_dev || !scratch_dev || !q_dev) { set_error("kate_division: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_poly_kate_division_dev");
    Ctx& c = ctx();
    (void)c;
    hipStream_t s = (hipStream_t)stream;
    uint4* t = (uint4*)scratch_dev;
    uint4* P = t + 2 * n;
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (field == TRH_FP) {
        hipLaunchKernelGGL((mul_pointwise_kernel<FpParams>), dim3(gb), dim3(256), 0, s, (const uint4*)a_dev, (const uint4*)pz_dev, t, n);
        TRH_TRY((prefix_scan_t<FpParams, OpAdd<FpParams>>(t, P, n, s)));
        hipLaunchKernelGGL((kate_finish_kernel<FpParams>), dim3(gb), dim3(256), 0, s, t, P, (const uint4*)pzinv_dev, (uint4*)q_dev, n);
    } else {
        hipLaunchKernelGGL((mul_pointwise_kernel<FqParams>), dim3(gb), dim3(256), 0, s, (const uint4*)a_dev, (const uint4*)pz_dev, t, n);
        TRH_TRY((prefix_scan_t<FqParams, OpAdd<FqParams>>(t, P, n, s)));
        hipLaunchKernelGGL((kate_finish_kernel<FqParams>), dim3(gb), dim3(256), 0, s, t, P, (const uint4*)pzinv_dev, (uint4*)q_dev, n);
    }
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

int trh_field_prefix_product_rows_dev(int field, const void* a_dev, void* out_dev, size_t n, size_t rows, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (n && rows && (!a_dev || !out_dev)) { set_error("prefix_product_rows: null pointer"); return TRH_EINVAL; }
    if (a_dev == out_dev) { set_error("prefix_product_rows: in-place operation is not supported"); return TRH_EINVAL; }
    if (!n || !rows) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_field_prefix_product_rows_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return prefix_scan_t<FpParams, OpMul<FpParams>>(a_dev, out_dev, n, (hipStream_t)stream, rows);
    return prefix_scan_t<FqParams, OpMul<FqParams>>(a_dev, out_dev, n, (hipStream_t)stream, rows);
}

int trh_field_prefix_sum_dev(int field, const void* a_dev, void* out_dev, size_t n, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (n && (!a_dev || !out_dev)) { set_error("prefix_sum: null pointer"); return TRH_EINVAL; }
    if (a_dev == out_dev) { set_error("prefix_sum: in-place operation is not supported"); return TRH_EINVAL; }
    if (!n) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_field_prefix_sum_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return prefix_scan_t<FpParams, OpAdd<FpParams>>(a_dev, out_dev, n, (hipStream_t)stream);
    return prefix_scan_t<FqParams, OpAdd<FqParams>>(a_dev, out_dev, n, (hipStream_t)stream);
}

}  // extern "C"
