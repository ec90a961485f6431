// halo2_proofs 0.2.0 `poly::EvaluationDomain` on device polynomials (src/poly/domain.rs; reached from
// keygen_* / create_proof, reference call sites /root/reference/src/test_utils.rs:23-25, 41-49;
// SURVEY.md section 8 row a5).  The domain constants are derived on the host from the pasta
// ROOT_OF_UNITY / ZETA exactly as `EvaluationDomain::new(j, k)` does; the transforms are libtrh's NTT
// plus the pointwise steps: x n^-1, the zeta-coset shift (fused here with the zero-padding of
// coeff_to_extended), and the division by the vanishing polynomial on the coset.
#include <string.h>

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

int check(const trh_domain* d, const void* a) {
    TRH_TRY(require_init());
    if (!d || !a) { set_error("domain: null pointer"); return TRH_EINVAL; }
    if (d->device != ctx().device) { set_error("domain: created on device %d, called from a context on device %d", d->device, ctx().device); return TRH_EINVAL; }
    return TRH_OK;
}

}  // namespace
}  // namespace trh

using namespace trh;

extern "C" {

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
    if (rc != TRH_OK) { delete d; return rc; }
    *out = d;
    return TRH_OK;
}
void trh_domain_destroy(trh_domain* d) {
    if (!d) return;
    if (d->d_tables) (void)hipFree(d->d_tables);
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

/* EvaluationDomain::lagrange_to_coeff: iFFT with omega^-1, then * 2^-k; batch polynomials of 2^k, in place */
int trh_domain_lagrange_to_coeff(trh_domain* d, void* a_dev, size_t batch, void* stream) {
    TRH_TRY(check(d, a_dev));
    TRH_ENTER(stream);
    Range range("trh_domain_lagrange_to_coeff");
    Ctx& c = ctx();
    (void)c;
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


/* ---- the same operations on HOST polynomials (one pointer per column), pipelined over PCIe (hostio.hip): what the Rust host's
 * EvaluationDomain calls become when create_proof keeps its polynomials in host memory ------------------------------------- */
static size_t group_for(size_t bytes_per_column) {
    size_t g = 1;
    while (g < 64 && g * bytes_per_column < ((size_t)8 << 20)) g <<= 1;  // a pipeline item is worth a few MiB of link time
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
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) TRH_TRY(stage_h2d(c, (char*)din + (j - it * group) * bytes, a[j], bytes, c.stage.us));
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
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) TRH_TRY(stage_h2d(c, (char*)din + (j - it * group) * in_b, coeff[j], in_b, c.stage.us));
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

/* h(X): [divide_by_vanishing_poly,] extended_to_coeff on one host polynomial of 2^extended_k values, in place */
int trh_domain_extended_to_coeff_host(trh_domain* d, uint64_t* a, int divide_by_vanishing_first) {
    TRH_TRY(check(d, a));
    TRH_ENTER(0);
    Range range("trh_domain_extended_to_coeff_host");
    Ctx& c = ctx();
    TRH_TRY(stage_begin(c));
    hipStream_t s = c.stage.cs;
    const size_t bytes = (size_t)32 << d->extended_k;
    TRH_TRY(c.io.ensure(bytes));
    TRH_TRY(stage_h2d(c, c.io.p, a, bytes, s));
    if (divide_by_vanishing_first) TRH_TRY(trh_domain_divide_by_vanishing_poly(d, c.io.p, 1, s));
    TRH_TRY(trh_domain_extended_to_coeff(d, c.io.p, 1, s));
    TRH_TRY(stage_d2h(c, a, c.io.p, bytes, s));
    return stage_end(c);
}

}  // extern "C"
