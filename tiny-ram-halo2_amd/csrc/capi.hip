// C ABI of libtrh.so (include/trh.h): argument checking, staging of host buffers, context.
#include <stdlib.h>
#include <string.h>

#include "ctx.h"

namespace trh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

Ctx& ctx() {
    static Ctx c;
    return c;
}

int require_init() {
    if (!ctx().inited) {
        set_error("trh_init() has not succeeded: no HIP device bound (libtrh has no CPU fallback)");
        return TRH_ENODEV;
    }
    return TRH_OK;
}

int field_scale_periodic(int field, void* a_dev, size_t rows, size_t row_len, size_t active_len, const void* factors_dev, u32 period, hipStream_t s);

namespace {

template <class F>
__global__ void __launch_bounds__(256) field_op_kernel(int op, const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> x, y = fe_zero<F>(), r;
    {
        uint4 lo = a[2 * i], hi = a[2 * i + 1];
        x = fe_load<F>(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
    }
    if (b) {
        uint4 lo = b[2 * i], hi = b[2 * i + 1];
        y = fe_load<F>(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
    }
    switch (op) {
        case 0: r = fe_add(x, y); break;
        case 1: r = fe_sub(x, y); break;
        case 2: r = fe_mul(x, y); break;
        case 3: r = fe_sqr(x); break;
        case 4: r = fe_neg(x); break;
        case 5: r = fe_inv(x); break;
        case 6: r = fe_to_mont(x); break;
        default: r = fe_from_mont(x); break;
    }
    u32 w[8];
    fe_store(r, w);
    out[2 * i] = make_uint4(w[0], w[1], w[2], w[3]);
    out[2 * i + 1] = make_uint4(w[4], w[5], w[6], w[7]);
}

template <class F>
__global__ void __launch_bounds__(64) point_op_kernel(int op, const JacobianMem* __restrict__ p, const void* __restrict__ q, JacobianMem* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<F> a = xyzz_from_jacobian(jac_load<F>(p[i])), r;
    if (op == 0) {
        r = xyzz_add(a, xyzz_from_jacobian(jac_load<F>(((const JacobianMem*)q)[i])));
    } else if (op == 1) {
        r = a;
        xyzz_madd(r, aff_load<F>(((const AffineMem*)q)[i]));
    } else {
        r = xyzz_dbl(a);
    }
    jac_store(jac_from_affine(xyzz_to_affine(r)), out[i]);
}

int check_curve(int curve) {
    if (curve != TRH_PALLAS && curve != TRH_VESTA) { set_error("unknown curve id %d", curve); return TRH_EINVAL; }
    return TRH_OK;
}
int check_field(int field) {
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    return TRH_OK;
}

// ---- cache of base sets seen by the host-pointer entry points ---------------------------------------------------
// `best_multiexp(coeffs, bases)` is called ~500 times per proof with the SAME `Params.g_lagrange` / `Params.g` slices.
// The plain two-function shim passes host pointers every time; re-uploading 64 B per base
// would cost as much as the MSM.  A set is recognised by (curve, pointer, length, fingerprint of 256 sampled points)
// and kept resident (8 sets, LRU); from its fourth use it also gets the fixed-base tables.  TRH_BASES_CACHE=0 turns
// the cache off (every call then uploads, as before).
struct BasesCacheEntry {
    int curve;
    const void* host;
    size_t n;
    uint64_t fp;
    trh_bases* h;
    uint64_t stamp;
    unsigned uses;
};
std::mutex g_cache_mu;
std::vector<BasesCacheEntry> g_cache;
uint64_t g_cache_stamp = 0;

uint64_t bases_fingerprint(const uint64_t* bases, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n;
    const size_t samples = n < 256 ? n : 256;
    for (size_t k = 0; k < samples; ++k) {
        const size_t i = samples == n ? k : (k * (n - 1)) / (samples - 1);  // includes the first and the last point
        for (int w = 0; w < 8; ++w) { h ^= bases[8 * i + w]; h *= 0x100000001b3ull; }
    }
    return h;
}

void bases_cache_clear() {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (BasesCacheEntry& e : g_cache) trh_bases_destroy(e.h);
    g_cache.clear();
}

int best_multiexp_host(int curve, const uint64_t* coeffs, const uint64_t* bases, size_t n, uint64_t* out) {
    TRH_TRY(require_init());
    if (!out || (n && (!coeffs || !bases))) { set_error("best_multiexp: null pointer"); return TRH_EINVAL; }
    if (n >= ((size_t)1 << 31)) { set_error("best_multiexp: n too large"); return TRH_EINVAL; }
    static const int cache_on = getenv("TRH_BASES_CACHE") ? atoi(getenv("TRH_BASES_CACHE")) : 1;
    if (cache_on && n >= 1024) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        const uint64_t fp = bases_fingerprint(bases, n);
        BasesCacheEntry* hit = nullptr;
        for (BasesCacheEntry& e : g_cache)
            if (e.curve == curve && e.host == (const void*)bases && e.n == n && e.fp == fp) { hit = &e; break; }
        if (!hit) {
            if (g_cache.size() >= 8) {
                size_t victim = 0;
                for (size_t i = 1; i < g_cache.size(); ++i) if (g_cache[i].stamp < g_cache[victim].stamp) victim = i;
                trh_bases_destroy(g_cache[victim].h);
                g_cache.erase(g_cache.begin() + victim);
            }
            trh_bases* h = nullptr;
            TRH_TRY(curve == TRH_PALLAS ? trh_bases_create_pallas(bases, n, &h) : trh_bases_create_vesta(bases, n, &h));
            g_cache.push_back(BasesCacheEntry{curve, bases, n, fp, h, 0, 0});
            hit = &g_cache.back();
        }
        hit->stamp = ++g_cache_stamp;
        if (++hit->uses == 4) (void)trh_bases_precompute(hit->h, 0);  // outside the supported range: stays on the per-window path
        return trh_msm(hit->h, 0, coeffs, n, 1, out);
    }
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    DevBuf bbuf;
    TRH_TRY(c.msm.scalars.ensure(n * 32 + 32));
    int rc = bbuf.ensure(n * 64 + 64);
    if (rc != TRH_OK) return rc;
    hipError_t e = hipSuccess;
    if (n) {
        e = hipMemcpy(c.msm.scalars.p, coeffs, n * 32, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(bbuf.p, bases, n * 64, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { bbuf.release(); set_error("best_multiexp: upload failed: %s", hipGetErrorString(e)); return TRH_EHIP; }
    rc = msm_enqueue(curve, bbuf.p, nullptr, c.msm.scalars.p, n, 1, n, 1, 0);
    if (rc == TRH_OK) rc = msm_finish(curve, 0, out, 1);
    bbuf.release();
    return rc;
}

int best_fft_host(int field, uint64_t* a, const uint64_t* omega, uint32_t log_n) {
    TRH_TRY(require_init());
    if (!a || !omega) { set_error("best_fft: null pointer"); return TRH_EINVAL; }
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (log_n > 27) { set_error("best_fft: log_n %u > 27 unsupported", log_n); return TRH_EINVAL; }
    const size_t bytes = (size_t)32 << log_n;
    TRH_TRY(c.io.ensure(bytes));
    TRH_HIP_TRY(hipMemcpy(c.io.p, a, bytes, hipMemcpyHostToDevice));
    TRH_TRY(ntt_device(field, c.io.p, log_n, omega, 1, 0));
    TRH_HIP_TRY(hipMemcpy(a, c.io.p, bytes, hipMemcpyDeviceToHost));
    return TRH_OK;
}

int bases_create(int curve, const uint64_t* xy, size_t n, trh_bases_t* out) {
    TRH_TRY(require_init());
    if (!out || (n && !xy)) { set_error("bases_create: null pointer"); return TRH_EINVAL; }
    trh_bases* b = new trh_bases{curve, nullptr, n, true};
    hipError_t e = hipMalloc(&b->d_xy, n * 64 + 64);
    if (e == hipSuccess && n) e = hipMemcpy(b->d_xy, xy, n * 64, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (b->d_xy) (void)hipFree(b->d_xy);
        delete b;
        set_error("bases_create: %s", hipGetErrorString(e));
        return TRH_EHIP;
    }
    *out = b;
    return TRH_OK;
}

}  // namespace
}  // namespace trh

using namespace trh;

extern "C" {

const char* trh_version(void) { return "trh 0.1.0 (gfx950)"; }
const char* trh_last_error(void) { return g_err; }

int trh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int trh_init(int device) {
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.inited) {
        if (c.device == device) return TRH_OK;
        set_error("trh_init: already bound to device %d", c.device);
        return TRH_EINVAL;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_error("trh_init: no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e)); return TRH_ENODEV; }
    if (device < 0 || device >= n) { set_error("trh_init: device %d out of range [0, %d)", device, n); return TRH_EINVAL; }
    TRH_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    TRH_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_error("trh_init: device is %s, libtrh is built for gfx950 only", prop.gcnArchName); return TRH_ENODEV; }
    c.device = device;
    c.inited = true;
    return TRH_OK;
}

void trh_shutdown(void) {
    bases_cache_clear();  // before the context lock: the cache lock is always taken first
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.inited) return;
    (void)hipDeviceSynchronize();
    msm_release();
    lookup_release();
    ntt_release_tables();
    for (DevBuf& d : ctx().ipa) d.release();
    c.io.release();
    c.factors.release();
    if (c.pinned_ring) { (void)hipHostFree(c.pinned_ring); c.pinned_ring = nullptr; c.pinned_slot = 0; }
    if (c.pinned_land) { (void)hipHostFree(c.pinned_land); c.pinned_land = nullptr; }
    c.pfft.release();
    c.scan.release(); c.scan2.release();
    c.inited = false;
    c.device = -1;
}

int trh_best_multiexp_pallas(const uint64_t* coeffs, const uint64_t* bases, size_t n, uint64_t out[12]) { return best_multiexp_host(TRH_PALLAS, coeffs, bases, n, out); }
int trh_best_multiexp_vesta(const uint64_t* coeffs, const uint64_t* bases, size_t n, uint64_t out[12]) { return best_multiexp_host(TRH_VESTA, coeffs, bases, n, out); }
int trh_best_fft_fp(uint64_t* a, const uint64_t omega[4], uint32_t log_n) { return best_fft_host(TRH_FP, a, omega, log_n); }
int trh_best_fft_fq(uint64_t* a, const uint64_t omega[4], uint32_t log_n) { return best_fft_host(TRH_FQ, a, omega, log_n); }

int trh_bases_create_pallas(const uint64_t* xy, size_t n, trh_bases_t* out) { return bases_create(TRH_PALLAS, xy, n, out); }
int trh_bases_create_vesta(const uint64_t* xy, size_t n, trh_bases_t* out) { return bases_create(TRH_VESTA, xy, n, out); }

int trh_bases_wrap_device(int curve, const void* xy_dev, size_t n, trh_bases_t* out) {
    TRH_TRY(require_init());
    TRH_TRY(check_curve(curve));
    if (!out || (n && !xy_dev)) { set_error("bases_wrap_device: null pointer"); return TRH_EINVAL; }
    *out = new trh_bases{curve, (void*)xy_dev, n, false};
    return TRH_OK;
}

int trh_bases_generate(int curve, uint64_t s0, uint64_t d, uint64_t first, size_t n, trh_bases_t* out) {
    TRH_TRY(require_init());
    TRH_TRY(check_curve(curve));
    if (!out) { set_error("bases_generate: null pointer"); return TRH_EINVAL; }
    trh_bases* b = new trh_bases{curve, nullptr, n, true};
    hipError_t e = hipMalloc(&b->d_xy, n * 64 + 64);
    if (e != hipSuccess) { delete b; set_error("bases_generate: %s", hipGetErrorString(e)); return TRH_ENOMEM; }
    int rc = bases_generate_device(curve, s0, d, first, n, b->d_xy, 0);
    if (rc == TRH_OK && hipStreamSynchronize(0) != hipSuccess) { set_error("bases_generate: kernel failed"); rc = TRH_EHIP; }
    if (rc != TRH_OK) { (void)hipFree(b->d_xy); delete b; return rc; }
    *out = b;
    return TRH_OK;
}

int trh_bases_download(trh_bases_t b, size_t offset, size_t n, uint64_t* xy_host) {
    TRH_TRY(require_init());
    if (!b || !xy_host || offset + n > b->n) { set_error("bases_download: bad range"); return TRH_EINVAL; }
    TRH_HIP_TRY(hipMemcpy(xy_host, (const char*)b->d_xy + offset * 64, n * 64, hipMemcpyDeviceToHost));
    return TRH_OK;
}
const void* trh_bases_device_ptr(trh_bases_t b) { return b ? b->d_xy : nullptr; }
size_t trh_bases_len(trh_bases_t b) { return b ? b->n : 0; }
void trh_bases_destroy(trh_bases_t b) {
    if (!b) return;
    if (b->owned && b->d_xy) (void)hipFree(b->d_xy);
    if (b->d_z) (void)hipFree(b->d_z);
    if (b->d_table) (void)hipFree(b->d_table);
    delete b;
}

// owned base sets are immutable: convert them to the lazy Montgomery domain once and keep the copy
static const void* lazy_bases(trh_bases_t b, size_t offset, hipStream_t s) {
    if (!b->owned || b->n == 0) return nullptr;
    if (!b->d_z) {
        void* z = nullptr;
        if (hipMalloc(&z, b->n * 64 + 64) != hipSuccess) return nullptr;  // fall back to per-call conversion
        if (msm_convert_bases(b->curve, b->d_xy, z, b->n, s) != TRH_OK || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(z); return nullptr; }
        b->d_z = z;
    }
    return (const char*)b->d_z + offset * 64;
}

// the fixed-base table covers the whole set: used for full-range MSMs unless a window width is forced
static const MsmFixedBase* fixed_base(trh_bases_t b, size_t offset, size_t n) {
    return (b->d_table && offset == 0 && n == b->n && ctx().window_override == 0) ? &b->fb : nullptr;
}

int trh_bases_precompute(trh_bases_t b, int window_bits) {
    TRH_TRY(require_init());
    if (!b) { set_error("bases_precompute: null handle"); return TRH_EINVAL; }
    if (!b->owned) { set_error("bases_precompute: wrapped device memory may change under the table; create an owned set"); return TRH_EINVAL; }
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (b->d_table) { (void)hipFree(b->d_table); b->d_table = nullptr; b->fb = MsmFixedBase{nullptr, 0, 0}; }
    if (b->n == 0) return TRH_OK;
    int cb = window_bits;
    if (cb == 0) {  // wide windows: the single reduction costs 2^(c-1) additions per MSM against n * ceil(255 / c) mixed adds
        int l = 0;
        while (((size_t)2 << l) <= b->n) ++l;
        cb = l - 2 < 6 ? 6 : l - 2 > 17 ? 17 : l - 2;  // measured at 2^18 (batch 32): c = 16 beats 15 / 17
    }
    if (!msm_fixed_base_fits(b->n, cb)) { set_error("bases_precompute: window width %d with %zu bases is outside the fixed-base range (W * n <= 2^24, c <= 18)", cb, b->n); return TRH_EINVAL; }
    const int W = msm_fixed_base_windows(cb);
    void* t = nullptr;
    hipError_t e = hipMalloc(&t, (size_t)W * b->n * 64 + 64);
    if (e != hipSuccess) { set_error("bases_precompute: hipMalloc(%zu): %s", (size_t)W * b->n * 64, hipGetErrorString(e)); return TRH_ENOMEM; }
    int rc = msm_build_table(b->curve, b->d_xy, b->n, cb, t, 0);
    if (rc == TRH_OK && hipStreamSynchronize(0) != hipSuccess) { set_error("bases_precompute: table kernel failed"); rc = TRH_EHIP; }
    if (rc != TRH_OK) { (void)hipFree(t); return rc; }
    b->d_table = t;
    b->fb = MsmFixedBase{t, cb, W};
    return TRH_OK;
}
int trh_bases_precomputed_window_bits(trh_bases_t b) { return b && b->d_table ? b->fb.c : 0; }

static int msm_args(trh_bases_t bases, size_t offset, const void* scalars, size_t n, size_t batch, void* out) {
    TRH_TRY(require_init());
    if (!bases || !out || (n && !scalars)) { set_error("msm: null pointer"); return TRH_EINVAL; }
    if (offset + n > bases->n || offset + n < offset) { set_error("msm: range [%zu, %zu) exceeds the %zu resident bases", offset, offset + n, bases->n); return TRH_EINVAL; }
    if (n >= ((size_t)1 << 31)) { set_error("msm: n too large"); return TRH_EINVAL; }
    if (batch == 0) { set_error("msm: batch == 0"); return TRH_EINVAL; }
    return TRH_OK;
}

int trh_msm(trh_bases_t bases, size_t offset, const uint64_t* scalars_host, size_t n, int mont, uint64_t out[12]) {
    TRH_TRY(msm_args(bases, offset, scalars_host, n, 1, out));
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    TRH_TRY(c.msm.scalars.ensure(n * 32 + 32));
    if (n) TRH_HIP_TRY(hipMemcpy(c.msm.scalars.p, scalars_host, n * 32, hipMemcpyHostToDevice));
    TRH_TRY(msm_enqueue(bases->curve, (const char*)bases->d_xy + offset * 64, lazy_bases(bases, offset, 0), c.msm.scalars.p, n, 1, n, mont, 0, fixed_base(bases, offset, n)));
    return msm_finish(bases->curve, 0, out, 1);
}

int trh_msm_dev_enqueue(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n, int mont, void* stream) {
    uint64_t dummy;
    TRH_TRY(msm_args(bases, offset, scalars_dev, n, 1, &dummy));
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    return msm_enqueue(bases->curve, (const char*)bases->d_xy + offset * 64, lazy_bases(bases, offset, (hipStream_t)stream), scalars_dev, n, 1, n, mont, (hipStream_t)stream, fixed_base(bases, offset, n));
}
int trh_msm_dev_finish(trh_bases_t bases, void* stream, uint64_t out[12]) {
    TRH_TRY(require_init());
    if (!bases || !out) { set_error("msm_dev_finish: null pointer"); return TRH_EINVAL; }
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    return msm_finish(bases->curve, (hipStream_t)stream, out, 1);
}
int trh_msm_dev(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n, int mont, void* stream, uint64_t out[12]) {
    TRH_TRY(msm_args(bases, offset, scalars_dev, n, 1, out));
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    TRH_TRY(msm_enqueue(bases->curve, (const char*)bases->d_xy + offset * 64, lazy_bases(bases, offset, (hipStream_t)stream), scalars_dev, n, 1, n, mont, (hipStream_t)stream, fixed_base(bases, offset, n)));
    return msm_finish(bases->curve, (hipStream_t)stream, out, 1);
}
int trh_msm_batch_dev(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n, size_t batch, int mont, void* stream, uint64_t* out) {
    TRH_TRY(msm_args(bases, offset, scalars_dev, n, batch, out));
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    TRH_TRY(msm_enqueue(bases->curve, (const char*)bases->d_xy + offset * 64, lazy_bases(bases, offset, (hipStream_t)stream), scalars_dev, n, batch, n, mont, (hipStream_t)stream, fixed_base(bases, offset, n)));
    return msm_finish(bases->curve, (hipStream_t)stream, out, batch);
}
/* Params::commit / commit_lagrange for `batch` polynomials resident on the device: item b is the MSM of
 * polys[b] (n scalars) || blinds[b] over the n + 1 bases of the handle (g or g_lagrange followed by w) */
int trh_commit_batch_dev(trh_bases_t bases, const void* polys_dev, size_t n, size_t batch, const uint64_t* blinds_host, void* stream, uint64_t* out) {
    TRH_TRY(msm_args(bases, 0, polys_dev, n + 1, batch, out));
    if (!blinds_host) { set_error("commit_batch: null blinds"); return TRH_EINVAL; }
    if (bases->n != n + 1) { set_error("commit_batch: the handle must hold n + 1 = %zu bases (g or g_lagrange followed by w), it holds %zu", n + 1, bases->n); return TRH_EINVAL; }
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    hipStream_t s = (hipStream_t)stream;
    TRH_TRY(c.msm.tails.ensure(batch * 32));
    TRH_HIP_TRY(hipMemcpyAsync(c.msm.tails.p, blinds_host, batch * 32, hipMemcpyHostToDevice, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));  // the caller's blinds may be reused
    TRH_TRY(msm_enqueue(bases->curve, bases->d_xy, lazy_bases(bases, 0, s), polys_dev, n + 1, batch, n, 1, s, fixed_base(bases, 0, n + 1), c.msm.tails.p));
    return msm_finish(bases->curve, s, out, batch);
}

int trh_msm_set_window_bits(int cbits) {
    if (cbits != 0 && (cbits < 2 || cbits > 18)) { set_error("window bits must be 0 or in [2, 18]"); return TRH_EINVAL; }
    ctx().window_override = cbits;
    return TRH_OK;
}

int trh_point_sum(int curve, const uint64_t* pts, size_t count, uint64_t out[12]) {
    TRH_TRY(check_curve(curve));
    if (!out || (count && !pts)) { set_error("point_sum: null pointer"); return TRH_EINVAL; }
    return point_sum_host(curve, pts, count, out);
}

int trh_ntt_dev(int field, void* a_dev, uint32_t log_n, const uint64_t omega[4], size_t batch, void* stream) {
    TRH_TRY(require_init());
    TRH_TRY(check_field(field));
    if (!a_dev || !omega) { set_error("ntt_dev: null pointer"); return TRH_EINVAL; }
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    return ntt_device(field, a_dev, log_n, omega, batch, (hipStream_t)stream);
}

int trh_field_scale_rows_dev(int field, void* a_dev, size_t rows, size_t row_len, size_t active_len, const uint64_t* factors, uint32_t period, void* stream) {
    TRH_TRY(require_init());
    TRH_TRY(check_field(field));
    if (!a_dev || !factors || period == 0 || period > 64 || active_len > row_len) { set_error("field_scale: bad arguments"); return TRH_EINVAL; }
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    // factors are staged in a small ring so that back-to-back calls on one stream do not overwrite
    // a table a queued kernel still reads
    TRH_TRY(c.factors.ensure(16 * 64 * 32));
    char* slot = (char*)c.factors.p + (size_t)(c.factor_slot++ & 15) * 64 * 32;
    TRH_HIP_TRY(hipMemcpyAsync(slot, factors, (size_t)period * 32, hipMemcpyHostToDevice, (hipStream_t)stream));
    TRH_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // `factors` is caller memory
    return field_scale_periodic(field, a_dev, rows, row_len, active_len, slot, period, (hipStream_t)stream);
}
int trh_field_scale_periodic_dev(int field, void* a_dev, size_t n, const uint64_t* factors, uint32_t period, void* stream) {
    return trh_field_scale_rows_dev(field, a_dev, 1, n, n, factors, period, stream);
}
int trh_field_scale_dev(int field, void* a_dev, size_t n, const uint64_t factor[4], void* stream) {
    return trh_field_scale_periodic_dev(field, a_dev, n, factor, 1, stream);
}

int trh_field_op_dev(int field, int op, const void* a, const void* b, void* out, size_t n, void* stream) {
    TRH_TRY(require_init());
    TRH_TRY(check_field(field));
    if (!n) return TRH_OK;
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (field == TRH_FP) hipLaunchKernelGGL((field_op_kernel<FpParams>), dim3(gb), dim3(256), 0, (hipStream_t)stream, op, (const uint4*)a, (const uint4*)b, (uint4*)out, n);
    else hipLaunchKernelGGL((field_op_kernel<FqParams>), dim3(gb), dim3(256), 0, (hipStream_t)stream, op, (const uint4*)a, (const uint4*)b, (uint4*)out, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}
int trh_point_op_dev(int curve, int op, const void* p, const void* q, void* out, size_t n, void* stream) {
    TRH_TRY(require_init());
    TRH_TRY(check_curve(curve));
    if (!n) return TRH_OK;
    const unsigned gb = (unsigned)((n + 63) / 64);
    if (curve == TRH_PALLAS) hipLaunchKernelGGL((point_op_kernel<FpParams>), dim3(gb), dim3(64), 0, (hipStream_t)stream, op, (const JacobianMem*)p, q, (JacobianMem*)out, n);
    else hipLaunchKernelGGL((point_op_kernel<FqParams>), dim3(gb), dim3(64), 0, (hipStream_t)stream, op, (const JacobianMem*)p, q, (JacobianMem*)out, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

int trh_malloc(void** dev, size_t bytes) {
    TRH_TRY(require_init());
    if (!dev) { set_error("trh_malloc: null pointer"); return TRH_EINVAL; }
    hipError_t e = hipMalloc(dev, bytes ? bytes : 16);
    if (e != hipSuccess) { set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return TRH_ENOMEM; }
    return TRH_OK;
}
int trh_free(void* dev) {
    TRH_TRY(require_init());
    TRH_HIP_TRY(hipFree(dev));
    return TRH_OK;
}
int trh_memcpy_h2d(void* dev, const void* host, size_t bytes) {
    TRH_TRY(require_init());
    TRH_HIP_TRY(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return TRH_OK;
}
int trh_memcpy_d2h(void* host, const void* dev, size_t bytes) {
    TRH_TRY(require_init());
    TRH_HIP_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return TRH_OK;
}
int trh_stream_synchronize(void* stream) {
    TRH_TRY(require_init());
    TRH_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return TRH_OK;
}

int trh_set_timing(int enabled) { ctx().timing = enabled; return TRH_OK; }
int trh_last_timing(trh_timing_t* out) {
    if (!out) { set_error("last_timing: null pointer"); return TRH_EINVAL; }
    *out = ctx().last;
    return TRH_OK;
}

}  // extern "C"
