// halo2_proofs 0.2.0 `poly::EvaluationDomain` on device polynomials (src/poly/domain.rs; reached from
// keygen_* / create_proof, reference call sites /root/reference/src/test_utils.rs:23-25, 41-49;
// SURVEY.md section 8 row a5).  The domain constants are derived on the host from the pasta
// ROOT_OF_UNITY / ZETA exactly as `EvaluationDomain::new(j, k)` does; the transforms are libtrh's NTT
// plus the pointwise steps: x n^-1, the zeta-coset shift (fused here with the zero-padding of
// coeff_to_extended), and the division by the vanishing polynomial on the coset.
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "ctx.h"

namespace trh {
int field_scale_periodic(int field, void* a_dev, size_t rows, size_t row_len, size_t active_len, const void* factors_dev, u32 period, hipStream_t s);
}

struct trh_domain {
    int field;
    uint32_t j, k, extended_k;
    trh::FeMem omega, omega_inv, extended_omega, extended_omega_inv, ifft_divisor, extended_ifft_divisor;
    trh::FeMem into_coset[3], from_coset[3];  // 1, zeta, zeta^2  /  1, zeta^2, zeta
    std::vector<trh::FeMem> t_inv;            // (X^n - 1)^-1 on the coset, period 2^(extended_k - k)
    void* d_tables = nullptr;                  // device copy: into_coset[3], from_coset[3], divisors[2], t_inv[...], then the lazy-form block
    trh::FeMem z_into[3], z_idiv, z_from_div[3];  // lazy Montgomery form (x 2^270) for the steps fused into the NTT passes
    int device = -1;                           // the tables live on this device
    // extended domain as coset blocks (trh_domain_coeff_to_extended_blocks): zeta * extended_omega^(r + q * 2^(extended_k - k)) =
    // (zeta * extended_omega^r) * omega^q -- block r is the size-2^k transform of the coefficients scaled by (zeta extended_omega^r)^j
    void* d_post_blocks = nullptr;   // [j - 1][2^k]: (zeta extended_omega^r)^-j * 2^-k, the way back from block r
    void* d_vinv = nullptr;          // two (j - 1) x (j - 1) matrices (canonical Montgomery): inverse of V[r][i] = c_r^i, c_r = (zeta extended_omega^r)^(2^k),
                                     // plain and with column r divided by (c_r - 1) (the vanishing polynomial's value on block r)
    // (zeta extended_omega^r)^j for r < n_blocks as a table of its own PER block count (the kernel addresses the table's planes by the row
    // count): built at first use, kept until trh_domain_destroy -- a handle is shared by contexts, and a table freed when the count
    // changed could still be read by another context's queued transform
    std::map<uint32_t, void*> pre_sub;
    std::mutex mu;                   // a domain is shared by the contexts of its device: the lazily built block tables are created under this lock
};

namespace trh {
namespace {

template <class F> FeMem mem(const Fe<F>& v) { FeMem m; fe_store(v, m); return m; }
template <class F> Fe<F> reg(const FeMem& m) { return fe_load<F>(m); }

template <class F>
void build_domain(trh_domain* d) {
    Fe<F> ext_omega = fe_load<F>(F::ROOT_OF_UNITY);
    for (uint32_t i = d->extended_k; i < 32; ++i) ext_omega = fe_sqr(ext_omega);
    Fe<F> omega = ext_omega;
    for (uint32_t i = d->k; i < d->extended_k; ++i) omega = fe_sqr(omega);
    const Fe<F> one = fe_one<F>(), zeta = fe_load<F>(F::ZETA), zeta2 = fe_sqr(zeta);
    d->omega = mem(omega); d->omega_inv = mem(fe_inv(omega));
    d->extended_omega = mem(ext_omega); d->extended_omega_inv = mem(fe_inv(ext_omega));
    const Fe<F> two_inv = fe_inv(fe_dbl(one));
    Fe<F> div = one;
    for (uint32_t i = 0; i < d->k; ++i) div = fe_mul(div, two_inv);
    d->ifft_divisor = mem(div);
    for (uint32_t i = d->k; i < d->extended_k; ++i) div = fe_mul(div, two_inv);
    d->extended_ifft_divisor = mem(div);
    d->into_coset[0] = mem(one); d->into_coset[1] = mem(zeta); d->into_coset[2] = mem(zeta2);
    d->from_coset[0] = mem(one); d->from_coset[1] = mem(zeta2); d->from_coset[2] = mem(zeta);
    // lazy form of the passes the factors ride in, f * 2^(256 + shift) = montmul(f * 2^256, 2^shift * 2^256): shift = 14 (2^270, the
    // unsigned 30-bit passes) or 5 (2^261, the signed 29-bit passes) -- ntt_lazy_shift()
    Fe<F> two14 = one;
    for (int i = 0; i < ntt_lazy_shift(); ++i) two14 = fe_dbl(two14);
    d->z_into[0] = mem(fe_mul(one, two14)); d->z_into[1] = mem(fe_mul(zeta, two14)); d->z_into[2] = mem(fe_mul(zeta2, two14));
    d->z_idiv = mem(fe_mul(reg<F>(d->ifft_divisor), two14));
    const Fe<F> ediv = reg<F>(d->extended_ifft_divisor);  // extended_to_coeff: 2^-extended_k and the inverse coset shift in one factor
    d->z_from_div[0] = mem(fe_mul(ediv, two14)); d->z_from_div[1] = mem(fe_mul(fe_mul(ediv, zeta2), two14)); d->z_from_div[2] = mem(fe_mul(fe_mul(ediv, zeta), two14));
    // t(X) = X^n - 1 on zeta * extended_omega^i: zeta^n * (extended_omega^n)^i - 1, inverted
    Fe<F> orig = zeta, step = ext_omega;
    for (uint32_t i = 0; i < d->k; ++i) { orig = fe_sqr(orig); step = fe_sqr(step); }
    Fe<F> cur = orig;
    const size_t period = (size_t)1 << (d->extended_k - d->k);
    d->t_inv.resize(period);
    for (size_t i = 0; i < period; ++i) {
        d->t_inv[i] = mem(fe_inv(fe_sub(cur, one)));
        cur = fe_mul(cur, step);
    }
}

// out[r][c] = c < n ? in[r][c] * f[c % 3] : 0   (zero-padding fused with distribute_powers_zeta)
template <class F>
__global__ void __launch_bounds__(256) pad_coset_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t rows, size_t n, size_t N, const uint4* __restrict__ f3) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * N) return;
    const size_t r = i / N, c = i - r * N;
    uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
    if (c < n) {
        const uint4* p = in + 2 * (r * n + c);
        lo = p[0]; hi = p[1];
        const u32 m3 = (u32)(c % 3);
        if (m3) {
            const uint4 fl = f3[2 * m3], fh = f3[2 * m3 + 1];
            const Fe<F> v = fe_mul(fe_load<F>(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w), fe_load<F>(fl.x, fl.y, fl.z, fl.w, fh.x, fh.y, fh.z, fh.w));
            u32 w[8];
            fe_store(v, w);
            lo = make_uint4(w[0], w[1], w[2], w[3]); hi = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
    out[2 * i] = lo; out[2 * i + 1] = hi;
}

// device table layout (FeMem units)
enum { T_INTO = 0, T_FROM = 3, T_IDIV = 6, T_EIDIV = 7, T_ZINTO = 8, T_ZIDIV = 11, T_ZFROMDIV = 12, T_TINV = 15 };

int upload_tables(trh_domain* d) {
    std::vector<FeMem> t(T_TINV + d->t_inv.size());
    for (int i = 0; i < 3; ++i) { t[T_INTO + i] = d->into_coset[i]; t[T_FROM + i] = d->from_coset[i]; }
    t[T_IDIV] = d->ifft_divisor; t[T_EIDIV] = d->extended_ifft_divisor;
    for (int i = 0; i < 3; ++i) { t[T_ZINTO + i] = d->z_into[i]; t[T_ZFROMDIV + i] = d->z_from_div[i]; }
    t[T_ZIDIV] = d->z_idiv;
    for (size_t i = 0; i < d->t_inv.size(); ++i) t[T_TINV + i] = d->t_inv[i];
    TRH_HIP_TRY(hipMalloc(&d->d_tables, t.size() * sizeof(FeMem)));
    const hipError_t e = hipMemcpy(d->d_tables, t.data(), t.size() * sizeof(FeMem), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d->d_tables);
        d->d_tables = nullptr;
        set_error("domain_create: table upload failed: %s", hipGetErrorString(e));
        return TRH_EHIP;
    }
    return TRH_OK;
}
const FeMem* tab(const trh_domain* d, int idx) { return (const FeMem*)d->d_tables + idx; }

// out[i][e] = sum_r M[i][r] * P[r][e]: the (j - 1) x (j - 1) solve that turns the blocks' residues mod (X^n - c_r) into the n-coefficient
// pieces of the quotient
template <class F>
__global__ void __launch_bounds__(256) block_combine_kernel(const uint4* __restrict__ p, uint4* __restrict__ out, const uint4* __restrict__ mat, u32 D, size_t n) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    Fe<F> v[8];
    for (u32 r = 0; r < D; ++r) { const uint4 lo = p[2 * (r * n + e)], hi = p[2 * (r * n + e) + 1]; v[r] = fe_load<F>(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w); }
    for (u32 i = 0; i < D; ++i) {
        Fe<F> acc = fe_zero<F>();
        for (u32 r = 0; r < D; ++r) {
            const uint4 lo = mat[2 * (i * D + r)], hi = mat[2 * (i * D + r) + 1];
            acc = fe_add(acc, fe_mul(v[r], fe_load<F>(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w)));
        }
        u32 w[8];
        fe_store(acc, w);
        out[2 * (i * n + e)] = make_uint4(w[0], w[1], w[2], w[3]);
        out[2 * (i * n + e) + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// block tables and the interpolation matrices, built at first use
template <class F>
int build_blocks(trh_domain* d, hipStream_t s) {
    std::lock_guard<std::mutex> lk(d->mu);
    if (d->d_post_blocks) return TRH_OK;
    const uint32_t nblk = 1u << (d->extended_k - d->k), D = d->j - 1;
    if (D > 8 || D > nblk) { set_error("domain blocks: quotient degree %u unsupported", D); return TRH_EINVAL; }
    const Fe<F> one = fe_one<F>(), zeta = fe_load<F>(F::ZETA), ext_omega = reg<F>(d->extended_omega);
    (void)nblk;
    void *post = nullptr, *vinv = nullptr;
    hipError_t e = hipMalloc(&post, ntt_block_table_bytes(D, d->k));
    if (e == hipSuccess) e = hipMalloc(&vinv, (size_t)2 * D * D * 32);
    if (e != hipSuccess) { if (post) (void)hipFree(post); set_error("domain blocks: %s", hipGetErrorString(e)); return TRH_ENOMEM; }
    const FeMem zi = mem(fe_inv(zeta)), wi = d->extended_omega_inv;
    int rc = ntt_block_table_build(d->field, post, D, d->k, (const u64*)&zi, (const u64*)&wi, (const u64*)&d->ifft_divisor, s);
    // c_r = (zeta ext_omega^r)^n; V[r][i] = c_r^i; invert by Gauss-Jordan over the field (D <= 8)
    std::vector<Fe<F>> c(D), a(D * D), inv(D * D);
    Fe<F> base = zeta;
    for (uint32_t r = 0; r < D; ++r) {
        Fe<F> t = base;
        for (uint32_t i = 0; i < d->k; ++i) t = fe_sqr(t);
        c[r] = t;
        base = fe_mul(base, ext_omega);
    }
    const Fe<F> zero = fe_zero<F>();
    for (uint32_t r = 0; r < D; ++r) { Fe<F> pw = one; for (uint32_t i = 0; i < D; ++i) { a[r * D + i] = pw; pw = fe_mul(pw, c[r]); inv[r * D + i] = r == i ? one : zero; } }
    auto is_zero = [](const Fe<F>& v) { u32 w[8]; fe_store(v, w); u32 o = 0; for (int i = 0; i < 8; ++i) o |= w[i]; return o == 0; };
    for (uint32_t col = 0; col < D && rc == TRH_OK; ++col) {
        uint32_t piv = col;
        while (piv < D && is_zero(a[piv * D + col])) ++piv;
        if (piv == D) { set_error("domain blocks: singular coset matrix"); rc = TRH_EINVAL; break; }
        for (uint32_t i = 0; i < D; ++i) { std::swap(a[piv * D + i], a[col * D + i]); std::swap(inv[piv * D + i], inv[col * D + i]); }
        const Fe<F> pinv = fe_inv(a[col * D + col]);
        for (uint32_t i = 0; i < D; ++i) { a[col * D + i] = fe_mul(a[col * D + i], pinv); inv[col * D + i] = fe_mul(inv[col * D + i], pinv); }
        for (uint32_t r = 0; r < D; ++r) {
            if (r == col) continue;
            const Fe<F> f = a[r * D + col];
            for (uint32_t i = 0; i < D; ++i) { a[r * D + i] = fe_sub(a[r * D + i], fe_mul(f, a[col * D + i])); inv[r * D + i] = fe_sub(inv[r * D + i], fe_mul(f, inv[col * D + i])); }
        }
    }
    if (rc == TRH_OK) {
        // inv = V^-1: h_i = sum_r inv[i][r] P_r.  Second matrix: the numerator's blocks are divided by t(X) = X^n - 1 = c_r - 1 on block r
        std::vector<FeMem> h(2 * D * D);
        for (uint32_t i = 0; i < D; ++i)
            for (uint32_t r = 0; r < D; ++r) {
                h[i * D + r] = mem(inv[i * D + r]);
                h[D * D + i * D + r] = mem(fe_mul(inv[i * D + r], fe_inv(fe_sub(c[r], one))));
            }
        if (hipMemcpy(vinv, h.data(), h.size() * 32, hipMemcpyHostToDevice) != hipSuccess) { set_error("domain blocks: matrix upload failed"); rc = TRH_EHIP; }
    }
    if (rc == TRH_OK && hipStreamSynchronize(s) != hipSuccess) { set_error("domain blocks: table kernel failed"); rc = TRH_EHIP; }
    if (rc != TRH_OK) { (void)hipFree(post); (void)hipFree(vinv); return rc; }
    d->d_post_blocks = post; d->d_vinv = vinv;
    return TRH_OK;
}

int check(const trh_domain* d, const void* a) {
    TRH_TRY(require_init());
    if (!d || !a) { set_error("domain: null pointer"); return TRH_EINVAL; }
    if (d->device != ctx().device) { set_error("domain: created on device %d, called from a context on device %d", d->device, ctx().device); return TRH_EINVAL; }
    return TRH_OK;
}

// (zeta extended_omega^r)^j for r < n_blocks, j < 2^k: one table per block count, built at first use (or by trh_domain_reserve)
int pre_block_table(trh_domain* d, uint32_t n_blocks, hipStream_t s, const void** out) {
    std::lock_guard<std::mutex> lk(d->mu);
    auto it = d->pre_sub.find(n_blocks);
    if (it == d->pre_sub.end()) {
        void* t = nullptr;
        TRH_HIP_TRY(hipMalloc(&t, ntt_block_table_bytes(n_blocks, d->k)));
        const FeMem zm = d->into_coset[1], onem = d->into_coset[0];
        int rc = ntt_block_table_build(d->field, t, n_blocks, d->k, (const u64*)&zm, (const u64*)&d->extended_omega, (const u64*)&onem, s);
        if (rc == TRH_OK && hipStreamSynchronize(s) != hipSuccess) { set_error("domain blocks: table kernel failed"); rc = TRH_EHIP; }  // complete before another context's stream may read it
        if (rc != TRH_OK) { (void)hipFree(t); return rc; }
        it = d->pre_sub.emplace(n_blocks, t).first;
    }
    *out = it->second;
    return TRH_OK;
}

}  // namespace
}  // namespace trh

using namespace trh;

extern "C" {

int trh_domain_reserve(trh_domain* d, size_t batch);
int trh_domain_create(int field, uint32_t j, uint32_t k, trh_domain** out) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!out || j < 2 || k > 27) { set_error("domain_create: bad arguments"); return TRH_EINVAL; }
    trh_domain* d = new trh_domain();
    d->field = field; d->j = j; d->k = k;
    uint32_t ek = k;
    while (((uint64_t)1 << ek) < ((uint64_t)1 << k) * (j - 1)) ++ek;
    if (ek > 27) { delete d; set_error("domain_create: extended_k %u > 27 unsupported", ek); return TRH_EINVAL; }
    d->extended_k = ek;
    if (field == TRH_FP) build_domain<FpParams>(d); else build_domain<FqParams>(d);
    TRH_ENTER(0);
    Range range("trh_domain_create");
    d->device = ctx().device;
    int rc = upload_tables(d);
    if (rc == TRH_OK) rc = trh_domain_reserve(d, 1);  // the transforms' tables now, not inside the first proof's steps
    if (rc != TRH_OK) { trh_domain_destroy(d); return rc; }
    *out = d;
    return TRH_OK;
}
void trh_domain_destroy(trh_domain* d) {
    if (!d) return;
    if (d->d_tables) (void)hipFree(d->d_tables);
    if (d->d_post_blocks) (void)hipFree(d->d_post_blocks);
    if (d->d_vinv) (void)hipFree(d->d_vinv);
    for (auto& kv : d->pre_sub) if (kv.second) (void)hipFree(kv.second);
    delete d;
}
uint32_t trh_domain_extended_k(trh_domain* d) { return d ? d->extended_k : 0; }
/* which: 0 omega, 1 omega_inv, 2 extended_omega, 3 extended_omega_inv, 4 ifft_divisor, 5 extended_ifft_divisor, 6 g_coset, 7 g_coset_inv */
int trh_domain_constant(trh_domain* d, int which, uint64_t out[4]) {
    if (!d || !out || which < 0 || which > 7) { set_error("domain_constant: bad arguments"); return TRH_EINVAL; }
    const FeMem* src[8] = {&d->omega, &d->omega_inv, &d->extended_omega, &d->extended_omega_inv, &d->ifft_divisor, &d->extended_ifft_divisor,
                           &d->into_coset[1], &d->into_coset[2]};
    memcpy(out, src[which], 32);
    return TRH_OK;
}

/* Tables and scratch of the per-column steps for batches of `batch` polynomials on the CALLING context (twiddle tables are per context,
 * the coset-block tables per domain): lagrange_to_coeff, coeff_to_extended_blocks with j - 1 blocks, blocks_to_quotient.  What the first
 * proof of a process would otherwise build and allocate inside its own steps (159.6 ms for the first proof against
 * 151.1 for the second); trh_domain_create calls it with batch = 1 (tables only). */
int trh_domain_reserve(trh_domain* d, size_t batch) {
    if (!d) { set_error("domain_reserve: null handle"); return TRH_EINVAL; }
    if (batch == 0) batch = 1;
    TRH_ENTER(0);
    Range range("trh_domain_reserve");
    if (!ntt_can_fuse(d->k)) return TRH_OK;  // small domains: canonical passes, nothing lasting to build
    hipStream_t s = nullptr;
    const uint32_t D = d->j - 1, nblk = 1u << (d->extended_k - d->k);
    TRH_TRY(ntt_prepare(d->field, d->k, (const u64*)&d->omega_inv, ntt_can_fold_scale(d->k) ? (const u64*)&d->ifft_divisor : nullptr, batch, 0, s));
    if (D <= 8 && D <= nblk) {
        TRH_TRY(d->field == TRH_FP ? build_blocks<FpParams>(d, s) : build_blocks<FqParams>(d, s));
        const void* pre = nullptr;
        TRH_TRY(pre_block_table(d, D, s, &pre));
        TRH_TRY(ntt_prepare(d->field, d->k, (const u64*)&d->omega, nullptr, batch * D, D, s));
        TRH_TRY(ntt_prepare(d->field, d->k, (const u64*)&d->omega_inv, nullptr, D, D, s));
    }
    TRH_HIP_TRY(hipStreamSynchronize(s));
    return TRH_OK;
}

/* EvaluationDomain::lagrange_to_coeff: iFFT with omega^-1, then * 2^-k; batch polynomials of 2^k, in place */
int trh_domain_lagrange_to_coeff(trh_domain* d, void* a_dev, size_t batch, void* stream) {
    TRH_TRY(check(d, a_dev));
    TRH_ENTER(stream);
    Range range("trh_domain_lagrange_to_coeff");
    Ctx& c = ctx();
    (void)c;
    if (ntt_can_fold_scale(d->k))  // x 2^-k inside the last pass's inter-pass twiddle table: no multiplication of its own
        return ntt_device(d->field, a_dev, d->k, (const u64*)&d->omega_inv, batch, (hipStream_t)stream, nullptr, (const u64*)&d->ifft_divisor);
    if (ntt_can_fuse(d->k)) {  // x 2^-k on the final store of the last pass
        NttFusion fu;
        fu.post = tab(d, T_ZIDIV); fu.post_period = 1;
        return ntt_device(d->field, a_dev, d->k, (const u64*)&d->omega_inv, batch, (hipStream_t)stream, &fu);
    }
    TRH_TRY(ntt_device(d->field, a_dev, d->k, (const u64*)&d->omega_inv, batch, (hipStream_t)stream));
    return field_scale_periodic(d->field, a_dev, 1, batch << d->k, batch << d->k, tab(d, T_IDIV), 1, (hipStream_t)stream);
}
/* EvaluationDomain::coeff_to_extended: zeta-coset shift + zero-pad (one kernel), FFT of size 2^extended_k.
 * coeff_dev: batch x 2^k, ext_dev: batch x 2^extended_k (output) */
int trh_domain_coeff_to_extended(trh_domain* d, const void* coeff_dev, void* ext_dev, size_t batch, void* stream) {
    TRH_TRY(check(d, coeff_dev));
    if (!ext_dev) { set_error("domain: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_domain_coeff_to_extended");
    Ctx& c = ctx();
    (void)c;
    const size_t n = (size_t)1 << d->k, N = (size_t)1 << d->extended_k, total = batch * N;
    if (ntt_can_fuse(d->extended_k)) {  // zero-padding and the zeta-coset shift happen on the loads of pass 0 (which also skips the stages that only see zeros)
        NttFusion fu;
        fu.in_dev = coeff_dev; fu.in_log = d->k;
        fu.pre = tab(d, T_ZINTO); fu.pre_period = 3;
        return ntt_device(d->field, ext_dev, d->extended_k, (const u64*)&d->extended_omega, batch, (hipStream_t)stream, &fu);
    }
    if (total) {
        const unsigned gb = (unsigned)((total + 255) / 256);
        if (d->field == TRH_FP) hipLaunchKernelGGL((pad_coset_kernel<FpParams>), dim3(gb), dim3(256), 0, (hipStream_t)stream, (const uint4*)coeff_dev, (uint4*)ext_dev, batch, n, N, (const uint4*)tab(d, T_INTO));
        else hipLaunchKernelGGL((pad_coset_kernel<FqParams>), dim3(gb), dim3(256), 0, (hipStream_t)stream, (const uint4*)coeff_dev, (uint4*)ext_dev, batch, n, N, (const uint4*)tab(d, T_INTO));
        TRH_HIP_TRY(hipGetLastError());
    }
    return ntt_device(d->field, ext_dev, d->extended_k, (const u64*)&d->extended_omega, batch, (hipStream_t)stream);
}
/* EvaluationDomain::extended_to_coeff: iFFT, * 2^-extended_k, inverse coset shift; in place on batch x 2^extended_k
 * (the caller truncates each polynomial to n * (j - 1) coefficients as the Rust code does) */
int trh_domain_extended_to_coeff(trh_domain* d, void* a_dev, size_t batch, void* stream) {
    TRH_TRY(check(d, a_dev));
    TRH_ENTER(stream);
    Range range("trh_domain_extended_to_coeff");
    Ctx& c = ctx();
    (void)c;
    const size_t N = (size_t)1 << d->extended_k;
    if (ntt_can_fuse(d->extended_k)) {  // 2^-extended_k * zeta^-(i mod 3) on the final store
        NttFusion fu;
        fu.post = tab(d, T_ZFROMDIV); fu.post_period = 3;
        return ntt_device(d->field, a_dev, d->extended_k, (const u64*)&d->extended_omega_inv, batch, (hipStream_t)stream, &fu);
    }
    TRH_TRY(ntt_device(d->field, a_dev, d->extended_k, (const u64*)&d->extended_omega_inv, batch, (hipStream_t)stream));
    TRH_TRY(field_scale_periodic(d->field, a_dev, 1, batch * N, batch * N, tab(d, T_EIDIV), 1, (hipStream_t)stream));
    return field_scale_periodic(d->field, a_dev, batch, N, N, tab(d, T_FROM), 3, (hipStream_t)stream);
}
/* EvaluationDomain::divide_by_vanishing_poly: a[i] *= t_inv[i % 2^(extended_k - k)] */
int trh_domain_divide_by_vanishing_poly(trh_domain* d, void* a_dev, size_t batch, void* stream) {
    TRH_TRY(check(d, a_dev));
    TRH_ENTER(stream);
    Range range("trh_domain_divide_by_vanishing_poly");
    Ctx& c = ctx();
    (void)c;
    const size_t N = (size_t)1 << d->extended_k;
    return field_scale_periodic(d->field, a_dev, batch, N, N, tab(d, T_TINV), (u32)d->t_inv.size(), (hipStream_t)stream);
}


/* ---- the extended domain as coset blocks ------------------------------------------------------------------------------------
 * The 2^extended_k points zeta * extended_omega^i split by i = q * 2^(extended_k - k) + r into 2^(extended_k - k) cosets
 * (zeta extended_omega^r) * omega^q of the size-2^k subgroup.  Block r of a polynomial a(X) of 2^k coefficients is the size-2^k
 * transform of a_j (zeta extended_omega^r)^j: two passes of a 2^18 transform instead of three of a zero-padded 2^21 one, and
 * the quotient h(X) of degree < (j - 1) 2^k only needs j - 1 of the blocks (5 of 8 for the reference's circuit: 5/8 of the
 * transforms and of the gate evaluation).  In this layout Rotation(1) is q + 1 inside a block (trh_expr_eval_blocks_dev).
 * ext_dev: batch x n_blocks x 2^k elements; block r of polynomial b at element ((b * n_blocks + r) << k); its entry q is entry
 * q * 2^(extended_k - k) + r of EvaluationDomain::coeff_to_extended.                                                             */
uint32_t trh_domain_quotient_blocks(trh_domain* d) { return d ? d->j - 1 : 0; }

int trh_domain_coeff_to_extended_blocks(trh_domain* d, const void* coeff_dev, void* ext_dev, size_t batch, uint32_t n_blocks, void* stream) {
    TRH_TRY(check(d, coeff_dev));
    if (!ext_dev || n_blocks == 0 || n_blocks > (1u << (d->extended_k - d->k))) { set_error("domain blocks: n_blocks out of range"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_domain_coeff_to_extended_blocks");
    Ctx& c = ctx();
    (void)c;
    hipStream_t s = (hipStream_t)stream;
    TRH_TRY(d->field == TRH_FP ? build_blocks<FpParams>(d, s) : build_blocks<FqParams>(d, s));
    const void* pre_tab = nullptr;
    TRH_TRY(pre_block_table(d, n_blocks, s, &pre_tab));
    if (ntt_can_fuse(d->k) && ntt_lazy_shift() == 5) {  // the scaling rides on the loads of pass 0
        NttFusion fu;
        fu.in_dev = coeff_dev; fu.in_log = d->k;
        fu.pre_blocks = pre_tab; fu.blocks = n_blocks;
        return ntt_device(d->field, ext_dev, d->k, (const u64*)&d->omega, batch * n_blocks, s, &fu);
    }
    // small domains: the same two steps unfused
    TRH_TRY(ntt_block_scale(d->field, coeff_dev, ext_dev, batch * n_blocks, n_blocks, d->k, pre_tab, false, s));
    return ntt_device(d->field, ext_dev, d->k, (const u64*)&d->omega, batch * n_blocks, s);
}

/* num_blocks_dev: the quotient's NUMERATOR (or, with divide_by_vanishing = 0, the quotient itself) on blocks 0 .. j - 2 of the
 * extended domain ((j - 1) x 2^k elements, overwritten); h_coeff_dev: (j - 1) x 2^k coefficients of h(X), piece i = coefficients
 * [i 2^k, (i + 1) 2^k) -- what extended_to_coeff(divide_by_vanishing_poly(.)) truncated to (j - 1) 2^k returns for a numerator that
 * the vanishing polynomial divides (create_proof's h(X)).  Block r yields h mod (X^n - c_r) = sum_i c_r^i h_i(X), an inverse
 * size-2^k coset transform per block and a (j - 1) x (j - 1) solve per coefficient.                                              */
int trh_domain_blocks_to_quotient(trh_domain* d, void* num_blocks_dev, void* h_coeff_dev, int divide_by_vanishing, void* stream) {
    TRH_TRY(check(d, num_blocks_dev));
    if (!h_coeff_dev) { set_error("domain: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_domain_blocks_to_quotient");
    hipStream_t s = (hipStream_t)stream;
    TRH_TRY(d->field == TRH_FP ? build_blocks<FpParams>(d, s) : build_blocks<FqParams>(d, s));
    const uint32_t D = d->j - 1;
    const size_t n = (size_t)1 << d->k;
    if (ntt_can_fuse(d->k) && ntt_lazy_shift() == 5) {
        NttFusion fu;
        fu.post_blocks = d->d_post_blocks; fu.blocks = D;
        TRH_TRY(ntt_device(d->field, num_blocks_dev, d->k, (const u64*)&d->omega_inv, D, s, &fu));
    } else {
        TRH_TRY(ntt_device(d->field, num_blocks_dev, d->k, (const u64*)&d->omega_inv, D, s));
        TRH_TRY(ntt_block_scale(d->field, num_blocks_dev, num_blocks_dev, D, D, d->k, d->d_post_blocks, true, s));
    }
    const uint4* mat = (const uint4*)d->d_vinv + (divide_by_vanishing ? (size_t)2 * D * D : 0);
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (d->field == TRH_FP) hipLaunchKernelGGL((block_combine_kernel<FpParams>), dim3(gb), dim3(256), 0, s, (const uint4*)num_blocks_dev, (uint4*)h_coeff_dev, mat, D, n);
    else hipLaunchKernelGGL((block_combine_kernel<FqParams>), dim3(gb), dim3(256), 0, s, (const uint4*)num_blocks_dev, (uint4*)h_coeff_dev, mat, D, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

/* ---- the same operations on HOST polynomials (one pointer per column), pipelined over PCIe (hostio.hip): what the Rust host's
 * EvaluationDomain calls become when create_proof keeps its polynomials in host memory ------------------------------------- */
static size_t group_for(size_t bytes_per_column) {
    size_t g = 1;
    while (g < 64 && g * bytes_per_column < ((size_t)32 << 20)) g <<= 1;  // a pipeline item is worth ~0.5 ms of link time (per-item hand-overs cost ~0.1 ms)
    return g;
}

int trh_domain_lagrange_to_coeff_host(trh_domain* d, uint64_t* const* a, size_t count) {
    TRH_TRY(check(d, a));
    for (size_t i = 0; i < count; ++i) if (!a[i]) { set_error("domain: column %zu is null", i); return TRH_EINVAL; }
    TRH_ENTER(0);
    Range range("trh_domain_lagrange_to_coeff_host");
    Ctx& c = ctx();
    const size_t bytes = (size_t)32 << d->k, group = group_for(bytes);
    HostPipe p;
    p.count = (count + group - 1) / group;
    p.in_bytes = group * bytes;
    p.in_place = true;
    p.upload = [&](size_t it, void* din) -> int {
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) TRH_TRY(stage_h2d(c, (char*)din + (j - it * group) * bytes, a[j], bytes, c.stage.us, true));
        return TRH_OK;
    };
    p.compute = [&](size_t it, void* din, void*, hipStream_t s) -> int {
        return trh_domain_lagrange_to_coeff(d, din, count - it * group < group ? count - it * group : group, s);
    };
    p.segments = [&](size_t it, const void* dout, std::vector<HostPipe::Seg>& out) {
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) out.push_back(HostPipe::Seg{a[j], (const char*)dout + (j - it * group) * bytes, bytes});
    };
    return host_pipeline(c, p);
}

/* coeff[i]: 2^k coefficients (read), ext[i]: 2^extended_k values (written): only the 2^k non-zero coefficients go up */
int trh_domain_coeff_to_extended_host(trh_domain* d, const uint64_t* const* coeff, uint64_t* const* ext, size_t count) {
    TRH_TRY(check(d, coeff));
    if (!ext) { set_error("domain: null pointer"); return TRH_EINVAL; }
    for (size_t i = 0; i < count; ++i) if (!coeff[i] || !ext[i]) { set_error("domain: column %zu is null", i); return TRH_EINVAL; }
    TRH_ENTER(0);
    Range range("trh_domain_coeff_to_extended_host");
    Ctx& c = ctx();
    const size_t in_b = (size_t)32 << d->k, out_b = (size_t)32 << d->extended_k, group = group_for(out_b);
    HostPipe p;
    p.count = (count + group - 1) / group;
    p.in_bytes = group * in_b;
    p.out_bytes = group * out_b;
    p.upload = [&](size_t it, void* din) -> int {
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) TRH_TRY(stage_h2d(c, (char*)din + (j - it * group) * in_b, coeff[j], in_b, c.stage.us, true));
        return TRH_OK;
    };
    p.compute = [&](size_t it, void* din, void* dout, hipStream_t s) -> int {
        return trh_domain_coeff_to_extended(d, din, dout, count - it * group < group ? count - it * group : group, s);
    };
    p.segments = [&](size_t it, const void* dout, std::vector<HostPipe::Seg>& out) {
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) out.push_back(HostPipe::Seg{ext[j], (const char*)dout + (j - it * group) * out_b, out_b});
    };
    return host_pipeline(c, p);
}

/* the coset-block form for a host that evaluates h(X) itself but has adopted the block layout: n_blocks x 2^k values per column come
 * down instead of 2^extended_k (5/8 of the bytes for the reference's circuit), and the quotient goes back up as (j - 1) x 2^k values */
int trh_domain_coeff_to_extended_blocks_host(trh_domain* d, const uint64_t* const* coeff, uint64_t* const* ext, size_t count, uint32_t n_blocks) {
    TRH_TRY(check(d, coeff));
    if (!ext || n_blocks == 0 || n_blocks > (1u << (d->extended_k - d->k))) { set_error("domain blocks: bad arguments"); return TRH_EINVAL; }
    for (size_t i = 0; i < count; ++i) if (!coeff[i] || !ext[i]) { set_error("domain: column %zu is null", i); return TRH_EINVAL; }
    TRH_ENTER(0);
    Range range("trh_domain_coeff_to_extended_blocks_host");
    Ctx& c = ctx();
    const size_t in_b = (size_t)32 << d->k, out_b = (size_t)n_blocks * in_b, group = group_for(out_b);
    HostPipe p;
    p.count = (count + group - 1) / group;
    p.in_bytes = group * in_b;
    p.out_bytes = group * out_b;
    p.upload = [&](size_t it, void* din) -> int {
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) TRH_TRY(stage_h2d(c, (char*)din + (j - it * group) * in_b, coeff[j], in_b, c.stage.us, true));
        return TRH_OK;
    };
    p.compute = [&](size_t it, void* din, void* dout, hipStream_t s) -> int {
        return trh_domain_coeff_to_extended_blocks(d, din, dout, count - it * group < group ? count - it * group : group, n_blocks, s);
    };
    p.segments = [&](size_t it, const void* dout, std::vector<HostPipe::Seg>& out) {
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) out.push_back(HostPipe::Seg{ext[j], (const char*)dout + (j - it * group) * out_b, out_b});
    };
    return host_pipeline(c, p);
}

/* num_blocks: (j - 1) x 2^k host values of the numerator on blocks 0 .. j - 2; h_coeff: (j - 1) x 2^k host coefficients (may alias) */
int trh_domain_blocks_to_quotient_host(trh_domain* d, const uint64_t* num_blocks, uint64_t* h_coeff, int divide_by_vanishing) {
    TRH_TRY(check(d, num_blocks));
    if (!h_coeff) { set_error("domain: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(0);
    Range range("trh_domain_blocks_to_quotient_host");
    Ctx& c = ctx();
    TRH_TRY(stage_begin(c));
    StageScope scope(c);
    hipStream_t s = c.stage.cs;
    const size_t bytes = (size_t)(d->j - 1) * ((size_t)32 << d->k);
    TRH_TRY(c.io.ensure(2 * bytes));
    TRH_TRY(stage_h2d(c, c.io.p, num_blocks, bytes, s));
    TRH_TRY(trh_domain_blocks_to_quotient(d, c.io.p, (char*)c.io.p + bytes, divide_by_vanishing, s));
    TRH_TRY(stage_d2h(c, h_coeff, (char*)c.io.p + bytes, bytes, s));
    return scope.finish();
}

/* h(X): [divide_by_vanishing_poly,] extended_to_coeff on one host polynomial of 2^extended_k values, in place */
int trh_domain_extended_to_coeff_host(trh_domain* d, uint64_t* a, int divide_by_vanishing_first) {
    TRH_TRY(check(d, a));
    TRH_ENTER(0);
    Range range("trh_domain_extended_to_coeff_host");
    Ctx& c = ctx();
    TRH_TRY(stage_begin(c));
    StageScope scope(c);
    hipStream_t s = c.stage.cs;
    const size_t bytes = (size_t)32 << d->extended_k;
    TRH_TRY(c.io.ensure(bytes));
    TRH_TRY(stage_h2d(c, c.io.p, a, bytes, s));
    if (divide_by_vanishing_first) TRH_TRY(trh_domain_divide_by_vanishing_poly(d, c.io.p, 1, s));
    TRH_TRY(trh_domain_extended_to_coeff(d, c.io.p, 1, s));
    TRH_TRY(stage_d2h(c, a, c.io.p, bytes, s));
    return scope.finish();
}

}  // extern "C"
