// C ABI of libtrh.so (include/trh.h): argument checking, staging of host buffers, contexts.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <thread>

#include "ctx.h"
#include "hosthelper.h"
#include "curve_q4.h"
#include "devpool.h"

#ifndef TRH_BUILD_ID
#define TRH_BUILD_ID "unknown"
#endif

struct trh_ctx : trh::Ctx {};

namespace trh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- options (ctx.h Options): environment once, then trh_set_option while no context exists ------------------------------
namespace {
Options g_opt;
std::once_flag g_opt_once;
std::mutex g_opt_mu;                 // serialises trh_set_option calls; readers need none (nothing writes while a context is alive)
std::atomic<int> g_live_ctx{0};      // contexts alive: while > 0 the options are fixed
struct OptField { const char* name; int kind; void* p; long lo, hi; };  // kind 0: int, 1: long
const OptField* opt_fields(size_t* count) {
    static const OptField f[] = {
        {"pool_mb", 1, &g_opt.pool_mb, 0, 1 << 20},          {"stage_slot_mb", 1, &g_opt.stage_slot_mb, 1, 256}, {"copy_threads", 0, &g_opt.copy_threads, -1, 64},
        {"bases_cache", 0, &g_opt.bases_cache, 0, 1},        {"force_no_peer", 0, &g_opt.force_no_peer, 0, 1},   {"ipa_fold", 0, &g_opt.ipa_fold, 0, 10},
        {"trace", 0, &g_opt.trace, 0, 3},                    {"msm_chunk_gb", 1, &g_opt.msm_chunk_gb, 1, 256},   {"sparse", 0, &g_opt.sparse, 0, 1},
        {"reduce_q4", 0, &g_opt.reduce_q4, 0, 1},            {"bin_sort", 0, &g_opt.bin_sort, 0, 1},             {"selftest", 0, &g_opt.selftest, 0, 1},
    };
    *count = sizeof(f) / sizeof(f[0]);
    return f;
}
bool opt_assign(const OptField& f, const char* value) {
    char* end = nullptr;
    const long v = strtol(value, &end, 10);
    if (end == value || *end != 0 || v < f.lo || v > f.hi) return false;
    if (f.kind == 0) *(int*)f.p = (int)v; else *(long*)f.p = v;
    return true;
}
void opt_load_env() {  // TRH_<NAME> for every field; a value out of range is ignored (the default stays)
    size_t n = 0;
    const OptField* f = opt_fields(&n);
    for (size_t i = 0; i < n; ++i) {
        char env[64] = "TRH_";
        size_t k = 4;
        for (const char* q = f[i].name; *q && k + 1 < sizeof(env); ++q) env[k++] = (char)(*q >= 'a' && *q <= 'z' ? *q - 32 : *q);
        env[k] = 0;
        if (const char* e = getenv(env)) (void)opt_assign(f[i], e);
    }
}
}  // namespace
const Options& opt() {
    std::call_once(g_opt_once, opt_load_env);
    return g_opt;
}

// ---- context registry ------------------------------------------------------------------------------------------------
static std::mutex g_reg_mu;
static Ctx* g_default = nullptr;            // trh_init / trh_init_multi
static std::vector<Ctx*> g_group;           // trh_init_multi: g_group[0] == g_default
static size_t g_shard_min = (size_t)1 << 20;  // smaller base sets stay on the default device
static int g_peer_ok = 1;                     // every pair of distinct group devices has peer access enabled
static thread_local Ctx* t_bound = nullptr;   // trh_ctx_set_current
static thread_local Ctx* t_active = nullptr;  // innermost TRH_ENTER

static Ctx* thread_ctx() { return t_bound ? t_bound : g_default; }

Ctx& ctx() {
    Ctx* c = t_active ? t_active : thread_ctx();
    if (!c) {  // unreachable through the C ABI (every entry point enters first); keep the failure loud
        fprintf(stderr, "libtrh: internal error: ctx() without a context\n");
        abort();
    }
    return *c;
}

int require_init() {
    Ctx* c = t_active ? t_active : thread_ctx();
    if (!c || !c->inited) {
        set_error("trh_init() has not succeeded: no HIP device bound (libtrh has no CPU fallback)");
        return TRH_ENODEV;
    }
    return TRH_OK;
}

int Enter::begin(hipStream_t stream, Ctx* explicit_ctx) {
    Ctx* cc = explicit_ctx ? explicit_ctx : thread_ctx();
    if (!cc || !cc->inited) {
        set_error("trh_init() has not succeeded: no HIP device bound (libtrh has no CPU fallback)");
        return TRH_ENODEV;
    }
    cc->mu.lock();
    locked = true;
    c = cc;
    prev_active = t_active;
    t_active = cc;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != cc->device) {  // hipSetDevice is per thread: a rayon worker never called trh_init
        prev_device = cur;
        TRH_HIP_TRY(hipSetDevice(cc->device));
    }
    if (prev_active != cc) {  // outermost entry into this context
        if (cc->last_stream_valid && cc->last_stream != stream) TRH_HIP_TRY(hipStreamWaitEvent(stream, cc->order_ev, 0));
        entered_stream = stream;
        outermost = true;
    }
    return TRH_OK;
}

Enter::~Enter() {
    if (!locked) return;
    if (outermost) {  // whatever this call enqueued is what the next stream entering the context has to wait for
        if (hipEventRecord(c->order_ev, entered_stream) == hipSuccess) { c->last_stream = entered_stream; c->last_stream_valid = true; }
        else c->last_stream_valid = false;
    }
    if (prev_device >= 0) (void)hipSetDevice(prev_device);
    t_active = prev_active;
    c->mu.unlock();
}

// ---- roctx ranges (rocprofv3 --marker-trace): resolved at first use, absent library = no-op -----------------------------
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
void roctx_resolve() {
    void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
    g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
    if (!g_roctx_push || !g_roctx_pop) { g_roctx_push = nullptr; g_roctx_pop = nullptr; }
}
}  // namespace
Range::Range(const char* name) {
    std::call_once(g_roctx_once, roctx_resolve);
    if (g_roctx_push) g_roctx_push(name);
}
Range::~Range() {
    if (g_roctx_pop) g_roctx_pop();
}

static int create_ctx(int device, Ctx** out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_error("no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e)); return TRH_ENODEV; }
    if (device < 0 || device >= n) { set_error("device %d out of range [0, %d)", device, n); return TRH_EINVAL; }
    hipDeviceProp_t prop;
    TRH_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_error("device %d is %s, libtrh is built for gfx950 only", device, prop.gcnArchName); return TRH_ENODEV; }
    int cur = -1;
    (void)hipGetDevice(&cur);
    TRH_HIP_TRY(hipSetDevice(device));
    trh_ctx* h = new trh_ctx();
    Ctx* c = h;
    c->device = device;
    hipError_t e1 = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    hipError_t e2 = hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming);
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    if (e1 != hipSuccess || e2 != hipSuccess) { delete h; set_error("context creation failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); return TRH_EHIP; }
    c->inited = true;
    g_live_ctx.fetch_add(1);
    *out = c;
    return TRH_OK;
}

namespace {
struct DevPool : DevPoolIndex {  // devpool.h: live / idle maps and the eviction order (tested on its own with made-up device ids)
    std::mutex mu;
};
DevPool g_pool;
size_t pool_cap() { return (size_t)opt().pool_mb << 20; }
size_t pool_round(size_t bytes) { return DevPoolIndex::round(bytes); }
void pool_release_idle() {  // g_pool.mu held
    for (auto& kv : g_pool.idle) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(kv.first.first);
        (void)hipFree(kv.second);
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    g_pool.idle.clear();
    g_pool.idle_bytes = 0;
}
}  // namespace

void pool_trim() {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    pool_release_idle();
}

static void destroy_ctx(Ctx* c) {
    if (!c) return;
    {
        Enter en;
        if (en.begin(nullptr, c) == TRH_OK) {
            (void)hipDeviceSynchronize();
            msm_release();
            lookup_release();
            ntt_release_tables();
            for (DevBuf& d : c->ipa) d.release();
            c->io.release();
            stage_release(*c);
            c->factors.release();
            if (c->pinned_ring) { (void)hipHostFree(c->pinned_ring); c->pinned_ring = nullptr; c->pinned_slot = 0; }
            if (c->pinned_land) { (void)hipHostFree(c->pinned_land); c->pinned_land = nullptr; }
            if (c->pinned_fold) { (void)hipHostFree(c->pinned_fold); c->pinned_fold = nullptr; c->pinned_fold_cap = 0; }
            c->pfft.release();
            c->scan.release(); c->scan2.release();
            c->last_stream_valid = false;
            en.outermost = false;  // nothing to order behind: the streams are drained
        }
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c->helper; c->helper = nullptr;
    if (c->order_ev) (void)hipEventDestroy(c->order_ev);
    c->inited = false;
    if (t_bound == c) t_bound = nullptr;
    delete static_cast<trh_ctx*>(c);
    g_live_ctx.fetch_sub(1);
}

int field_scale_periodic(int field, void* a_dev, size_t rows, size_t row_len, size_t active_len, const void* factors_dev, u32 period, hipStream_t s);

// selftest.hip: the known-answer test of the device arithmetic, once per device and process, on the context just created (made current
// for the calling thread while it runs).  On a mismatch the caller destroys the context and returns TRH_ESELFTEST.
int selftest_run();
static std::mutex g_selftest_mu;
static std::set<int> g_selftest_done;
static int selftest_once(Ctx* c) {
    if (!opt().selftest) return TRH_OK;
    std::lock_guard<std::mutex> lk(g_selftest_mu);
    if (g_selftest_done.count(c->device)) return TRH_OK;
    Ctx* prev = t_bound;
    t_bound = c;
    const int rc = selftest_run();
    t_bound = prev;
    if (rc == TRH_OK) g_selftest_done.insert(c->device);
    return rc;
}
// create_ctx + self-test; a context that fails it never reaches the caller
static int create_checked_ctx(int device, Ctx** out) {
    Ctx* c = nullptr;
    TRH_TRY(create_ctx(device, &c));
    const int rc = selftest_once(c);
    if (rc != TRH_OK) {
        char msg[512];
        snprintf(msg, sizeof(msg), "%s", g_err);
        destroy_ctx(c);
        set_error("%s", msg);
        return rc;
    }
    *out = c;
    return TRH_OK;
}

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

// ops 3 / 4: the same addition / doubling through the quad-lane arithmetic of curve_q4.h (four lanes per pair: lane q carries coordinate q in
// the lazy domain), so that the golden vectors and the edge cases (identity operands, P + P, P - P) test it directly
template <class F>
__global__ void __launch_bounds__(64) point_op_q4_kernel(int op, const JacobianMem* __restrict__ p, const void* __restrict__ q_in, JacobianMem* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int q = threadIdx.x & 3;
    if (i >= n) return;  // n is padded to whole quads by construction: the four lanes of a pair share i
    auto coord = [&](const XYZZz<F>& v) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.zz : v.zzz; };
    const Fy<F> a = coord(xyzzz_from_canonical(xyzz_from_jacobian(jac_load<F>(p[i]))));
    Fy<F> r;
    if (op == 3) r = q4_add(a, coord(xyzzz_from_canonical(xyzz_from_jacobian(jac_load<F>(((const JacobianMem*)q_in)[i])))), q);
    else r = q4_dbl(a, q);
    XYZZz<F> full;
    full.x = q4_perm<F, 0, 0, 0, 0>(r); full.y = q4_perm<F, 1, 1, 1, 1>(r); full.zz = q4_perm<F, 2, 2, 2, 2>(r); full.zzz = q4_perm<F, 3, 3, 3, 3>(r);
    if (q == 0) jac_store(jac_from_affine(xyzz_to_affine(xyzzz_to_canonical(full))), out[i]);
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
// would cost as much as the MSM.  OPT-IN (TRH_BASES_CACHE=1; the explicit form is trh_bases_create_*): a set is
// recognised by (curve, pointer, length, FNV-1a hash of ALL n points -- recomputed on every call, so a base changed in
// place is never served from the stale copy) and only becomes resident at its SECOND identical sighting, which keeps
// the once-per-proof IPA slices out of the 8-entry LRU; from its fourth use a set also gets the fixed-base tables.
struct BasesCacheEntry {
    int curve;
    const void* host;
    size_t n;
    uint64_t fp;
    trh_bases* h;
    uint64_t stamp;
    unsigned uses;
};
struct BasesCandidate { int curve; const void* host; size_t n; uint64_t fp; };
std::mutex g_cache_mu;
std::vector<BasesCacheEntry> g_cache;
BasesCandidate g_seen[16];
unsigned g_seen_next = 0;
uint64_t g_cache_stamp = 0;

uint64_t bases_fingerprint(const uint64_t* bases, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n;
    for (size_t k = 0; k < n * 8; ++k) { h ^= bases[k]; h *= 0x100000001b3ull; }
    return h;
}

void bases_cache_clear() {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (BasesCacheEntry& e : g_cache) trh_bases_destroy(e.h);
    g_cache.clear();
    for (BasesCandidate& c : g_seen) c = BasesCandidate{0, nullptr, 0, 0};
}

int sharded_create(int curve, const uint64_t* xy_host, uint64_t s0, uint64_t d, uint64_t first, size_t n, trh_bases_t* out);
int msm_sharded(trh_bases* B, size_t offset, const void* scalars, bool scalars_on_host, size_t n, int mont, hipStream_t caller_stream, uint64_t* out);

const void* lazy_bases(trh_bases_t b, size_t offset, hipStream_t s);
const MsmFixedBase* fixed_base(trh_bases_t b, size_t offset, size_t n);

// a synchronous entry point leaves no MSM "in flight" behind, whatever went wrong between its enqueue and its finish
// (otherwise every later call on the context would answer TRH_EBUSY)
struct PendingGuard {
    Ctx& c;
    bool armed = true;
    ~PendingGuard() {
        if (!armed || c.msm.pending_curve < 0) return;
        (void)hipStreamSynchronize(c.msm.pending_stream);
        c.msm.pending_curve = -1;
        c.msm.tile_sum_valid = false;
    }
};

// One MSM with the scalars (and, for best_multiexp, the bases) in HOST memory.  The sum over pairs is cut into ranges: range
// t + 1 crosses PCIe (stage_h2d on the upload stream, two device buffers) while range t is computed; the range points are added
// on the host.  With the bases on the host the call is bound by the link (96 B per pair: 2^24 pairs = 1.6 GB = 28 ms at 57 GB/s
// against 17 ms of arithmetic), so the ranges are small (2^20 pairs) and only the last one's arithmetic is exposed; with resident
// bases (32 B per pair) it is bound by the arithmetic and the ranges are large (2^22), only the first upload is exposed.
int msm_host_tiled(int curve, const uint64_t* coeffs, const uint64_t* bases_host, trh_bases* res, size_t offset, size_t n, int mont, uint64_t* out) {
    Ctx& c = ctx();
    TRH_TRY(stage_begin(c));
    StageScope scope(c);
    Stage& st = c.stage;
    // range boundaries
    std::vector<size_t> cut(1, 0);
    if (bases_host && n > ((size_t)1 << 21)) {  // equal ranges
        const size_t want = (size_t)1 << 20, nt = (n + want - 1) / want, len = (n + nt - 1) / nt;
        for (size_t o = len; o < n; o += len) cut.push_back(o);
    } else if (!bases_host && n > ((size_t)3 << 21)) {
        // growing ranges: 2^21, 2^22, then the rest -- the first upload is short, every later one hides under the range before it
        // (32 B per pair cross the link ~2x faster than they are multiplied), and most pairs run as one large MSM at the full rate
        cut.push_back((size_t)1 << 21);
        cut.push_back((size_t)3 << 21);
    }
    cut.push_back(n);
    const size_t ntiles = cut.size() - 1;
    uint64_t acc[24];
    memset(acc, 0, sizeof(acc));
    // option trace, bit 0: where the call's time goes (microseconds since here, stderr)
    const bool tr_on = (opt().trace & 1) != 0;
    const auto tr_t0 = std::chrono::steady_clock::now();
    auto tr_mark = [&](const char* what) {
        if (tr_on) fprintf(stderr, "[trh msm] %9.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr_t0).count(), what);
    };
    for (size_t t = 0; t < ntiles; ++t) {  // both device buffers at their final size before anything is in flight (a growing DevBuf frees)
        size_t longest = 0;
        for (size_t u = t & 1; u < ntiles; u += 2) longest = cut[u + 1] - cut[u] > longest ? cut[u + 1] - cut[u] : longest;
        if (t < 2) { TRH_TRY(st.ring_in[t].ensure(longest * 32 + 32)); if (bases_host) TRH_TRY(st.ring_out[t].ensure(longest * 64 + 64)); }
    }
    for (size_t t = 0; t < ntiles; ++t) {
        const size_t slot = t & 1, off = cut[t], cur = cut[t + 1] - cut[t];
        // (only the call's first bytes gate anything: every later range streams behind the one before it, full slots throughout)
        // (zero_elide: a witness column is zero on three quarters of its rows -- the chunks of the transfer that are zero throughout become a
        //  device-side memset instead of a copy into the pinned ring and a DMA; a chunk of random scalars answers "not zero" at its first word)
        TRH_TRY(stage_h2d(c, st.ring_in[slot].p, coeffs + 4 * off, cur * 32, st.us, t > 0, true, ntiles > 1 || bases_host != nullptr));
        if (bases_host) {
            TRH_TRY(stage_h2d(c, st.ring_out[slot].p, bases_host + 8 * off, cur * 64, st.us, true));
        }
        TRH_HIP_TRY(hipEventRecord(st.ev_up[slot], st.us));
        tr_mark("scalars handed to the upload stream");
        if (t > 0) {  // the previous range finished under this upload
            TRH_TRY(msm_finish(curve, st.cs, acc + 12, 1));
            TRH_TRY(point_sum_host(curve, acc, 2, acc));
        }
        TRH_HIP_TRY(hipStreamWaitEvent(st.cs, st.ev_up[slot], 0));
        const void* bdev = bases_host ? st.ring_out[slot].p : (const void*)((const char*)res->d_xy + (offset + off) * 64);
        const void* bz = bases_host ? nullptr : lazy_bases(res, offset + off, st.cs);
        // the fixed-base table covers whole sets of at most 2^24 / W pairs, which never split on their own (the growing ranges start above
        // 3 * 2^21 pairs); only a forced range length (TRH_HOST_TILE_LOG, a test switch) sends a tabled set down the per-window path
        const MsmFixedBase* fb = (!bases_host && ntiles == 1) ? fixed_base(res, offset, n) : nullptr;
        TRH_TRY(msm_enqueue(curve, bdev, bz, st.ring_in[slot].p, cur, 1, cur, mont, st.cs, fb));
        tr_mark("MSM enqueued");
    }
    TRH_TRY(msm_finish(curve, st.cs, acc + 12, 1));
    tr_mark("MSM finished");
    if (ntiles > 1) TRH_TRY(point_sum_host(curve, acc, 2, acc));
    memcpy(out, ntiles > 1 ? acc : acc + 12, 96);
    return scope.finish();
}

int best_multiexp_host(int curve, const uint64_t* coeffs, const uint64_t* bases, size_t n, uint64_t* out) {
    TRH_TRY(require_init());
    Range range(curve == TRH_PALLAS ? "trh_best_multiexp_pallas" : "trh_best_multiexp_vesta");
    if (!out || (n && (!coeffs || !bases))) { set_error("best_multiexp: null pointer"); return TRH_EINVAL; }
    if (n >= ((size_t)1 << 31)) { set_error("best_multiexp: n too large"); return TRH_EINVAL; }
    if (n == 0) { memset(out, 0, 96); return TRH_OK; }  // the empty sum: the identity (an empty Rust slice may carry any pointer)
    if (opt().bases_cache && n >= 1024) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        const uint64_t fp = bases_fingerprint(bases, n);
        BasesCacheEntry* hit = nullptr;
        for (BasesCacheEntry& e : g_cache)
            if (e.curve == curve && e.host == (const void*)bases && e.n == n && e.fp == fp) { hit = &e; break; }
        if (!hit) {
            bool seen = false;
            for (const BasesCandidate& c : g_seen) if (c.host == (const void*)bases && c.curve == curve && c.n == n && c.fp == fp) seen = true;
            if (!seen) g_seen[g_seen_next++ % 16] = BasesCandidate{curve, bases, n, fp};
            else {
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
        }
        if (hit) {
            hit->stamp = ++g_cache_stamp;
            if (++hit->uses == 4 && hit->h->shards.empty()) (void)trh_bases_precompute(hit->h, 0);  // outside the supported range: stays on the per-window path
            return trh_msm(hit->h, 0, coeffs, n, 1, out);
        }
    }
    if (g_group.size() > 1 && n >= g_shard_min && thread_ctx() == g_default) {  // device group: every GPU takes a range of the pairs
        trh_bases* h = nullptr;
        TRH_TRY(sharded_create(curve, bases, 0, 0, 0, n, &h));
        const int rc = msm_sharded(h, 0, coeffs, true, n, 1, nullptr, out);
        trh_bases_destroy(h);
        return rc;
    }
    TRH_ENTER(0);
    PendingGuard guard{ctx()};
    if (ctx().msm.pending_curve >= 0) { guard.armed = false; set_error("best_multiexp: this context has an enqueued MSM that was not finished"); return TRH_EBUSY; }
    return msm_host_tiled(curve, coeffs, bases, nullptr, 0, n, 1, out);
}

// one resident set on the entered context's device: uploaded from the host, or generated (xy == null)
int bases_create_local(int curve, const uint64_t* xy, uint64_t s0, uint64_t d, uint64_t first, size_t n, trh_bases_t* out) {
    trh_bases* b = new trh_bases{curve, nullptr, n, true};
    b->owner = &ctx();
    hipError_t e = hipMalloc(&b->d_xy, n * 64 + 64);
    if (e == hipSuccess && n && xy) e = hipMemcpy(b->d_xy, xy, n * 64, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (b->d_xy) (void)hipFree(b->d_xy);
        delete b;
        set_error("bases_create: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? TRH_ENOMEM : TRH_EHIP;
    }
    if (!xy && n) {
        int rc = bases_generate_device(curve, s0, d, first, n, b->d_xy, 0);
        if (rc == TRH_OK && hipStreamSynchronize(0) != hipSuccess) { set_error("bases_generate: kernel failed"); rc = TRH_EHIP; }
        if (rc != TRH_OK) { (void)hipFree(b->d_xy); delete b; return rc; }
    }
    *out = b;
    return TRH_OK;
}

// ---- range-sharded base sets over the device group (trh_init_multi) ---------------------------------------------------
// Shard g of G owns the pairs [g * per, (g + 1) * per) (per = ceil(n / G)) on the group's g-th context -- the partition
// best_multiexp itself makes per rayon thread.  An MSM enqueues one local Pippenger per shard on that context's own stream from
// the calling host thread, then collects the G partial points with plain device-to-host copies into pinned memory and adds them
// on the host (point_sum_host).  No RCCL here: EC addition is not a reduce op, the payload is 96 B per GPU, the result is
// consumed by the host (transcript), and everything happens inside ONE process -- a collective would only add a rendezvous.
// (bench.py's process-per-GPU harness does use RCCL's all_gather for the same 96-byte partials: there the ranks are processes.)
int sharded_create(int curve, const uint64_t* xy_host, uint64_t s0, uint64_t d, uint64_t first, size_t n, trh_bases_t* out) {
    const size_t G = g_group.size();
    std::unique_ptr<trh_bases> B(new trh_bases{curve, nullptr, n, true});
    B->owner = g_default;
    const size_t per = (n + G - 1) / G;
    B->shard_off.push_back(0);
    for (size_t g = 0; g < G; ++g) {
        const size_t lo = g * per < n ? g * per : n, hi = lo + per < n ? lo + per : n;
        trh_bases* sh = nullptr;
        int rc;
        {
            Enter en;
            rc = en.begin(nullptr, g_group[g]);
            if (rc == TRH_OK) rc = bases_create_local(curve, xy_host ? xy_host + 8 * lo : nullptr, s0, d, first + lo, hi - lo, &sh);
        }
        if (rc != TRH_OK) { for (trh_bases* q : B->shards) trh_bases_destroy(q); return rc; }
        B->shards.push_back(sh);
        B->shard_off.push_back(hi);
    }
    *out = B.release();
    return TRH_OK;
}

// TRH_FORCE_NO_PEER=1: the group behaves as if no pair of devices had peer access -- trh_init_multi enables none, and device-resident
// scalars reach EVERY shard (the one on the source device included) through the pinned-host hand-over that a box without peer access
// takes.  For exercising that path on a one-GPU box ({0, 0} groups).
bool force_no_peer() { return opt().force_no_peer != 0; }

int msm_sharded(trh_bases* B, size_t offset, const void* scalars, bool scalars_on_host, size_t n, int mont, hipStream_t caller_stream, uint64_t* out) {
    Range range("trh_msm[sharded]");
    const size_t G = B->shards.size();
    int src_device = -1;
    hipEvent_t ready = nullptr;
    if (!scalars_on_host && n) {  // device scalars live where the caller's context is: the shards copy their range over xGMI once the caller's stream got there
        Enter en;
        TRH_TRY(en.begin(caller_stream));
        src_device = ctx().device;
        ready = ctx().order_ev;  // recorded below on caller_stream; the context stays locked until the shard streams were told to wait for it
        TRH_HIP_TRY(hipEventRecord(ready, caller_stream));
        for (size_t g = 0; g < G; ++g) TRH_HIP_TRY(hipStreamWaitEvent(B->shards[g]->owner->own_stream, ready, 0));
        en.outermost = false;  // order_ev was just recorded by hand
        ctx().last_stream = caller_stream; ctx().last_stream_valid = true;
    }
    // lock the shard contexts in group order (every sharded call takes them in this order), enqueue everywhere, then collect.
    // The scopes nest, so they are left in REVERSE order on every path (each ~Enter restores the active context and the HIP device
    // it found: front-to-back destruction left the calling thread on shard G - 2's device); a shard whose MSM was
    // enqueued but not collected (an error in between) gets its in-flight state cleared, or the context would answer TRH_EBUSY forever
    struct HeldScopes {
        std::vector<std::unique_ptr<Enter>> v;
        explicit HeldScopes(size_t n) : v(n) {}
        ~HeldScopes() {
            while (!v.empty()) {
                if (v.back() && v.back()->c && v.back()->c->msm.pending_curve >= 0) {
                    (void)hipStreamSynchronize(v.back()->c->own_stream);
                    v.back()->c->msm.pending_curve = -1;
                }
                v.pop_back();
            }
        }
        std::unique_ptr<Enter>& operator[](size_t i) { return v[i]; }
    } held(G);
    std::vector<uint64_t> partial(12 * G, 0);
    std::vector<char> active(G, 0);
    struct Range1 { size_t cnt = 0, local = 0; const char* src = nullptr; };
    std::vector<Range1> rg(G);
    for (size_t g = 0; g < G; ++g) {
        const size_t lo = B->shard_off[g] > offset ? B->shard_off[g] : offset;
        const size_t hi = B->shard_off[g + 1] < offset + n ? B->shard_off[g + 1] : offset + n;
        if (hi <= lo) continue;
        Ctx* sc = B->shards[g]->owner;
        held[g].reset(new Enter());
        TRH_TRY(held[g]->begin(sc->own_stream, sc));
        rg[g].cnt = hi - lo; rg[g].local = lo - B->shard_off[g];
        rg[g].src = (const char*)scalars + (lo - offset) * 32;
        TRH_TRY(sc->msm.scalars.ensure(rg[g].cnt * 32 + 32));
        if (scalars_on_host) TRH_TRY(stage_ensure(*sc));  // rings and copy threads exist before the uploader threads start
        active[g] = 1;
    }
    // Host scalars: ONE uploader thread per shard, each feeding its device through that context's own pinned ring and copy threads (the
    // G links of a node run in parallel; one thread's memcpy stream -- 60-85 GB/s -- would be the limit of eight 57 GB/s links: a 2^26
    // MSM over 8 GPUs spent ~30 ms uploading against ~9 ms of arithmetic).  The calling thread holds every shard context's
    // lock and touches none of their stages meanwhile; it joins uploader g before it enqueues shard g's MSM, so the MSM of shard g starts
    // while the ranges of the later shards are still crossing.  Caller memory that is page-locked goes to the DMA engines directly.
    struct Upload { std::thread th; int rc = TRH_OK; std::string err; };
    std::vector<Upload> up(scalars_on_host ? G : 0);
    size_t n_active = 0;
    for (size_t g = 0; g < G; ++g) n_active += active[g] ? 1 : 0;
    const bool threaded = scalars_on_host && n_active > 1;
    auto join_all = [&] { for (Upload& u : up) if (u.th.joinable()) u.th.join(); };
    if (threaded) {
        for (size_t g = 0; g < G; ++g) {
            if (!active[g]) continue;
            Ctx* sc = B->shards[g]->owner;
            Upload* u = &up[g];
            const Range1 r = rg[g];
            u->th = std::thread([sc, u, r] {
                if (hipSetDevice(sc->device) != hipSuccess) { u->rc = TRH_EHIP; u->err = "hipSetDevice failed in an upload thread"; return; }
                u->rc = stage_h2d(*sc, sc->msm.scalars.p, r.src, r.cnt * 32, sc->own_stream);
                if (u->rc != TRH_OK) u->err = trh_last_error();
            });
        }
    }
    for (size_t g = 0; g < G; ++g) {
        if (!active[g]) continue;
        trh_bases* sh = B->shards[g];
        Ctx* sc = sh->owner;
        Enter en;
        int rc = en.begin(sc->own_stream, sc);  // re-entrant: the context is held above, this makes it the active one (and its device current) again
        if (rc == TRH_OK) {
            if (threaded) {
                up[g].th.join();
                if (up[g].rc != TRH_OK) { set_error("%s", up[g].err.c_str()); rc = up[g].rc; }
            } else if (scalars_on_host) {
                rc = stage_h2d(*sc, sc->msm.scalars.p, rg[g].src, rg[g].cnt * 32, sc->own_stream);
            } else if (force_no_peer() || (!g_peer_ok && src_device != sc->device)) {
                // no peer access in the group (or forced): through the destination context's pinned ring, stream-ordered on both devices
                rc = stage_d2d_via_host(*sc, sc->msm.scalars.p, sc->own_stream, rg[g].src, src_device, caller_stream, rg[g].cnt * 32);
            } else if (src_device == sc->device) {
                if (hipMemcpyAsync(sc->msm.scalars.p, rg[g].src, rg[g].cnt * 32, hipMemcpyDeviceToDevice, sc->own_stream) != hipSuccess) { set_error("msm[sharded]: device copy failed"); rc = TRH_EHIP; }
            } else {
                if (hipMemcpyPeerAsync(sc->msm.scalars.p, sc->device, rg[g].src, src_device, rg[g].cnt * 32, sc->own_stream) != hipSuccess) { set_error("msm[sharded]: peer copy failed: %s", hipGetErrorString(hipGetLastError())); rc = TRH_EHIP; }
            }
        }
        if (rc == TRH_OK) {
            // no sparse vote on a shard: the sampler ends in a host synchronisation behind this shard's upload / hand-over, which
            // would hold back the enqueues of the later shards; a range-sharded MSM is a full-size one
            // (its own flag: dense_hint also selects the quad-lane combine, which is for callers that vouch for full-size scalars -- ADVICE r05)
            const bool keep = sc->msm.no_sparse_vote;
            sc->msm.no_sparse_vote = true;
            rc = msm_enqueue(B->curve, (const char*)sh->d_xy + rg[g].local * 64, lazy_bases(sh, rg[g].local, sc->own_stream), sc->msm.scalars.p, rg[g].cnt, 1, rg[g].cnt, mont, sc->own_stream,
                             fixed_base(sh, rg[g].local, rg[g].cnt));
            sc->msm.no_sparse_vote = keep;
        }
        if (rc != TRH_OK) { join_all(); return rc; }
    }
    size_t cntp = 0;
    for (size_t g = 0; g < G; ++g) {
        if (!active[g]) continue;
        Ctx* sc = B->shards[g]->owner;
        Enter en;
        TRH_TRY(en.begin(sc->own_stream, sc));  // re-entrant: the context is still held above, this makes it the active one again
        TRH_TRY(msm_finish(B->curve, sc->own_stream, partial.data() + 12 * cntp, 1));
        ++cntp;
    }
    return point_sum_host(B->curve, partial.data(), cntp, out);
}

}  // namespace
}  // namespace trh

using namespace trh;

extern "C" {

const char* trh_version(void) { return "trh 0.2.0 (gfx950, build " TRH_BUILD_ID ")"; }
const char* trh_last_error(void) { return g_err; }

int trh_set_option(const char* name, const char* value) {
    if (!name || !value) { set_error("trh_set_option: null pointer"); return TRH_EINVAL; }
    std::lock_guard<std::mutex> lk(g_opt_mu);
    (void)opt();  // the environment first: an explicit call overrides it
    if (g_live_ctx.load() > 0) { set_error("trh_set_option(%s): options are fixed while a context exists (call it before trh_init, or after trh_shutdown)", name); return TRH_EBUSY; }
    size_t n = 0;
    const OptField* f = opt_fields(&n);
    for (size_t i = 0; i < n; ++i)
        if (strcmp(f[i].name, name) == 0) {
            if (!opt_assign(f[i], value)) { set_error("trh_set_option(%s): value '%s' is not an integer in [%ld, %ld]", name, value, f[i].lo, f[i].hi); return TRH_EINVAL; }
            return TRH_OK;
        }
    set_error("trh_set_option: unknown option '%s'", name);
    return TRH_EINVAL;
}
int trh_get_option(const char* name, long* value) {
    if (!name || !value) { set_error("trh_get_option: null pointer"); return TRH_EINVAL; }
    (void)opt();
    size_t n = 0;
    const OptField* f = opt_fields(&n);
    for (size_t i = 0; i < n; ++i)
        if (strcmp(f[i].name, name) == 0) { *value = f[i].kind == 0 ? (long)*(int*)f[i].p : *(long*)f[i].p; return TRH_OK; }
    set_error("trh_get_option: unknown option '%s'", name);
    return TRH_EINVAL;
}

int trh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int trh_init(int device) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_default) {
        if (g_default->device == device) return TRH_OK;
        set_error("trh_init: already bound to device %d", g_default->device);
        return TRH_EINVAL;
    }
    Ctx* c = nullptr;
    int rc = create_checked_ctx(device, &c);
    if (rc != TRH_OK) {
        char msg[400];
        snprintf(msg, sizeof(msg), "%s", g_err);
        set_error("trh_init: %s", msg);
        return rc;
    }
    g_default = c;
    g_group.assign(1, c);
    return TRH_OK;
}

int trh_init_multi(const int* devices, int n_devices) {
    if (!devices || n_devices < 1 || n_devices > 64) { set_error("trh_init_multi: bad arguments"); return TRH_EINVAL; }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_default) {
        bool same = g_group.size() == (size_t)n_devices;
        for (int i = 0; same && i < n_devices; ++i) same = g_group[i]->device == devices[i];
        if (same) return TRH_OK;
        // after a plain trh_init(devices[0]) the group grows around the existing default context
        if (g_group.size() != 1 || g_default->device != devices[0]) {
            set_error("trh_init_multi: already initialised with another device list (trh_shutdown first)");
            return TRH_EINVAL;
        }
    }
    std::vector<Ctx*> made;
    if (g_default) made.push_back(g_default);
    for (int i = (int)made.size(); i < n_devices; ++i) {  // the same device may be listed more than once: two lanes on one GPU
        Ctx* c = nullptr;
        const int rc = create_checked_ctx(devices[i], &c);
        if (rc != TRH_OK) {
            for (Ctx* m : made) if (m != g_default) destroy_ctx(m);
            char msg[400];
            snprintf(msg, sizeof(msg), "%s", g_err);
            set_error("trh_init_multi: %s", msg);
            return rc;
        }
        made.push_back(c);
    }
    // peer access for the hand-over of device-resident scalars.  Not fatal when it cannot be had (msm_sharded then hands the
    // ranges over through pinned host memory, stage_d2d_via_host), but not silent either: trh_group_peer_access() reports it and trh_last_error() names the first pair
    g_peer_ok = 1;
    if (force_no_peer()) {
        g_peer_ok = 0;
        set_error("trh_init_multi: TRH_FORCE_NO_PEER=1: no peer access enabled; device-resident scalars are handed over through pinned host memory");
    }
    for (int i = 0; i < n_devices && !force_no_peer(); ++i)
        for (int j = 0; j < n_devices; ++j)
            if (devices[i] != devices[j]) {
                int can = 0;
                hipError_t e = hipDeviceCanAccessPeer(&can, devices[i], devices[j]);
                if (e == hipSuccess && can) {
                    int cur = -1;
                    (void)hipGetDevice(&cur);
                    e = hipSetDevice(devices[i]);
                    if (e == hipSuccess) {
                        e = hipDeviceEnablePeerAccess(devices[j], 0);
                        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
                    }
                    if (cur >= 0) (void)hipSetDevice(cur);
                }
                if (e != hipSuccess || !can) {
                    if (g_peer_ok) set_error("trh_init_multi: no peer access from device %d to device %d (%s); device-resident scalars will be staged through the host",
                                             devices[i], devices[j], e != hipSuccess ? hipGetErrorString(e) : "hipDeviceCanAccessPeer says no");
                    (void)hipGetLastError();
                    g_peer_ok = 0;
                }
            }
    g_group = made;
    g_default = made[0];
    return TRH_OK;
}
int trh_group_size(void) { return (int)g_group.size(); }
int trh_group_peer_access(void) { return g_peer_ok; }
int trh_set_shard_min(size_t n_pairs) { g_shard_min = n_pairs ? n_pairs : 1; return TRH_OK; }

void trh_shutdown(void) {
    bases_cache_clear();  // before the registry lock: destroying a cached set enters its context
    { std::lock_guard<std::mutex> pl(g_pool.mu); pool_release_idle(); }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (Ctx* c : g_group) destroy_ctx(c);
    g_group.clear();
    g_default = nullptr;
}

int trh_ctx_create(int device, trh_ctx_t* out) {
    if (!out) { set_error("ctx_create: null pointer"); return TRH_EINVAL; }
    Ctx* c = nullptr;
    TRH_TRY(create_checked_ctx(device, &c));
    *out = static_cast<trh_ctx*>(c);
    return TRH_OK;
}
void trh_ctx_destroy(trh_ctx_t c) { destroy_ctx(c); }
int trh_ctx_set_current(trh_ctx_t c) {
    if (c && !c->inited) { set_error("ctx_set_current: destroyed context"); return TRH_EINVAL; }
    t_bound = c;
    return TRH_OK;
}
void* trh_ctx_stream(trh_ctx_t c) {
    Ctx* cc = c ? (Ctx*)c : thread_ctx();
    return cc ? (void*)cc->own_stream : nullptr;
}
int trh_ctx_device(trh_ctx_t c) {
    Ctx* cc = c ? (Ctx*)c : thread_ctx();
    return cc ? cc->device : -1;
}

int trh_best_multiexp_pallas(const uint64_t* coeffs, const uint64_t* bases, size_t n, uint64_t out[12]) { return best_multiexp_host(TRH_PALLAS, coeffs, bases, n, out); }
int trh_best_multiexp_vesta(const uint64_t* coeffs, const uint64_t* bases, size_t n, uint64_t out[12]) { return best_multiexp_host(TRH_VESTA, coeffs, bases, n, out); }
int trh_best_fft_fp(uint64_t* a, const uint64_t omega[4], uint32_t log_n) { return best_fft_host(TRH_FP, a, omega, log_n); }
int trh_best_fft_fq(uint64_t* a, const uint64_t omega[4], uint32_t log_n) { return best_fft_host(TRH_FQ, a, omega, log_n); }

static int bases_create(int curve, const uint64_t* xy, size_t n, trh_bases_t* out) {
    TRH_TRY(require_init());
    if (!out || (n && !xy)) { set_error("bases_create: null pointer"); return TRH_EINVAL; }
    if (g_group.size() > 1 && n >= g_shard_min && thread_ctx() == g_default) return sharded_create(curve, xy, 0, 0, 0, n, out);
    TRH_ENTER(0);
    return bases_create_local(curve, xy, 0, 0, 0, n, out);
}
int trh_bases_create_pallas(const uint64_t* xy, size_t n, trh_bases_t* out) { return bases_create(TRH_PALLAS, xy, n, out); }
int trh_bases_create_vesta(const uint64_t* xy, size_t n, trh_bases_t* out) { return bases_create(TRH_VESTA, xy, n, out); }

int trh_bases_wrap_device(int curve, const void* xy_dev, size_t n, trh_bases_t* out) {
    TRH_TRY(check_curve(curve));
    if (!out || (n && !xy_dev)) { set_error("bases_wrap_device: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(0);
    trh_bases* b = new trh_bases{curve, (void*)xy_dev, n, false};
    b->owner = &ctx();
    *out = b;
    return TRH_OK;
}

int trh_bases_generate(int curve, uint64_t s0, uint64_t d, uint64_t first, size_t n, trh_bases_t* out) {
    TRH_TRY(require_init());
    TRH_TRY(check_curve(curve));
    if (!out) { set_error("bases_generate: null pointer"); return TRH_EINVAL; }
    if (g_group.size() > 1 && n >= g_shard_min && thread_ctx() == g_default) return sharded_create(curve, nullptr, s0, d, first, n, out);
    TRH_ENTER(0);
    return bases_create_local(curve, nullptr, s0, d, first, n, out);
}

int trh_bases_download(trh_bases_t b, size_t offset, size_t n, uint64_t* xy_host) {
    if (!b || !xy_host || offset + n > b->n) { set_error("bases_download: bad range"); return TRH_EINVAL; }
    if (!b->shards.empty()) {
        for (size_t g = 0; g < b->shards.size(); ++g) {
            const size_t lo = b->shard_off[g] > offset ? b->shard_off[g] : offset;
            const size_t hi = b->shard_off[g + 1] < offset + n ? b->shard_off[g + 1] : offset + n;
            if (hi > lo) TRH_TRY(trh_bases_download(b->shards[g], lo - b->shard_off[g], hi - lo, xy_host + 8 * (lo - offset)));
        }
        return TRH_OK;
    }
    TRH_ENTER_CTX(0, b->owner);
    TRH_HIP_TRY(hipMemcpy(xy_host, (const char*)b->d_xy + offset * 64, n * 64, hipMemcpyDeviceToHost));
    return TRH_OK;
}
const void* trh_bases_device_ptr(trh_bases_t b) { return b ? b->d_xy : nullptr; }
size_t trh_bases_len(trh_bases_t b) { return b ? b->n : 0; }
int trh_bases_shards(trh_bases_t b) { return b ? (b->shards.empty() ? 1 : (int)b->shards.size()) : 0; }
void trh_bases_destroy(trh_bases_t b) {
    if (!b) return;
    for (trh_bases* sh : b->shards) trh_bases_destroy(sh);
    if (b->owned && b->d_xy) (void)hipFree(b->d_xy);
    if (b->d_z) (void)hipFree(b->d_z);
    if (b->d_table) (void)hipFree(b->d_table);
    delete b;
}

}  // extern "C"

namespace trh {
namespace {
// owned base sets are immutable: convert them to the lazy Montgomery domain once and keep the copy
const void* lazy_bases(trh_bases_t b, size_t offset, hipStream_t s) {
    if (!b->owned || b->n == 0) return nullptr;
    std::lock_guard<std::mutex> lk(b->mu);
    if (!b->d_z) {
        void* z = nullptr;
        if (hipMalloc(&z, b->n * ZREC + ZREC) != hipSuccess) return nullptr;  // fall back to per-call conversion
        if (msm_convert_bases(b->curve, b->d_xy, z, b->n, s) != TRH_OK || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(z); return nullptr; }
        b->d_z = z;
    }
    return (const char*)b->d_z + offset * ZREC;
}

// the fixed-base table covers the whole set: used for full-range MSMs unless a window width is forced
const MsmFixedBase* fixed_base(trh_bases_t b, size_t offset, size_t n) {
    return (b->d_table && offset == 0 && n == b->n && ctx().window_override == 0) ? &b->fb : nullptr;
}

int msm_args(trh_bases_t bases, size_t offset, const void* scalars, size_t n, size_t batch, void* out) {
    TRH_TRY(require_init());
    if (!bases || !out || (n && !scalars)) { set_error("msm: null pointer"); return TRH_EINVAL; }
    if (offset + n > bases->n || offset + n < offset) { set_error("msm: range [%zu, %zu) exceeds the %zu resident bases", offset, offset + n, bases->n); return TRH_EINVAL; }
    if (n >= ((size_t)1 << 31)) { set_error("msm: n too large"); return TRH_EINVAL; }
    if (batch == 0) { set_error("msm: batch == 0"); return TRH_EINVAL; }
    return TRH_OK;
}
int single_device(trh_bases_t bases, const char* what) {
    if (!bases->shards.empty()) { set_error("%s: needs a base set on ONE device (this handle is range-sharded over %zu)", what, bases->shards.size()); return TRH_EINVAL; }
    if (bases->owner && bases->owner->device != ctx().device) { set_error("%s: the base set lives on device %d, the calling context on device %d", what, bases->owner->device, ctx().device); return TRH_EINVAL; }
    return TRH_OK;
}
}  // namespace
}  // namespace trh

extern "C" {

int trh_bases_precompute(trh_bases_t b, int window_bits) {
    if (!b) { set_error("bases_precompute: null handle"); return TRH_EINVAL; }
    if (!b->owned) { set_error("bases_precompute: wrapped device memory may change under the table; create an owned set"); return TRH_EINVAL; }
    if (!b->shards.empty()) { set_error("bases_precompute: range-sharded sets keep the per-window path"); return TRH_EINVAL; }
    TRH_ENTER_CTX(0, b->owner);
    Range range("trh_bases_precompute");
    std::lock_guard<std::mutex> lk(b->mu);  // (an MSM that is already running over the old table from another context is the caller's to exclude: trh.h)
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
    hipError_t e = hipMalloc(&t, (size_t)W * b->n * ZREC + ZREC);
    if (e != hipSuccess) { set_error("bases_precompute: hipMalloc(%zu): %s", (size_t)W * b->n * ZREC, hipGetErrorString(e)); return TRH_ENOMEM; }
    int rc = msm_build_table(b->curve, b->d_xy, b->n, cb, t, 0);
    if (rc == TRH_OK && hipStreamSynchronize(0) != hipSuccess) { set_error("bases_precompute: table kernel failed"); rc = TRH_EHIP; }
    if (rc != TRH_OK) { (void)hipFree(t); return rc; }
    b->d_table = t;
    b->fb = MsmFixedBase{t, cb, W};
    return TRH_OK;
}
int trh_bases_precomputed_window_bits(trh_bases_t b) { return b && b->d_table ? b->fb.c : 0; }

int trh_bases_reserve(trh_bases_t b, size_t n, size_t batch) {
    if (!b) { set_error("bases_reserve: null handle"); return TRH_EINVAL; }
    if (n == 0 || n > b->n || batch == 0) { set_error("bases_reserve: n = %zu, batch = %zu over a set of %zu bases", n, batch, b->n); return TRH_EINVAL; }
    if (!b->shards.empty()) return TRH_OK;  // every shard's context sizes itself at its first MSM
    TRH_ENTER(0);
    Range range("trh_bases_reserve");
    TRH_TRY(single_device(b, "bases_reserve"));
    Ctx& c = ctx();
    if (c.msm.pending_curve >= 0) { set_error("bases_reserve: this context has an enqueued MSM that was not finished"); return TRH_EBUSY; }
    TRH_TRY(c.msm.tails.ensure(batch * 32));
    c.msm.reserve_only = true;
    const int rc = msm_enqueue(b->curve, b->d_xy, lazy_bases(b, 0, nullptr), b->d_xy /* never read */, n, batch, n, 1, nullptr, fixed_base(b, 0, n), batch > 1 ? c.msm.tails.p : nullptr);
    c.msm.reserve_only = false;
    TRH_TRY(rc);
    // the shape of an IPA opening's round MSMs -- two rows over the whole of a tabled set of 2^k + 2 points (g || w || u): the opening's own buffers too
    if (batch == 2 && n == b->n && b->d_table && n > 2 && ((n - 2) & (n - 3)) == 0) {
        uint32_t k = 0;
        while (((size_t)1 << k) < n - 2) ++k;
        TRH_TRY(ipa_reserve(b->curve, b, k));
    }
    return TRH_OK;
}

int trh_msm(trh_bases_t bases, size_t offset, const uint64_t* scalars_host, size_t n, int mont, uint64_t out[12]) {
    TRH_TRY(msm_args(bases, offset, scalars_host, n, 1, out));
    if (!bases->shards.empty()) return msm_sharded(bases, offset, scalars_host, true, n, mont, nullptr, out);
    TRH_ENTER(0);
    Range range("trh_msm");
    TRH_TRY(single_device(bases, "msm"));
    PendingGuard guard{ctx()};
    if (ctx().msm.pending_curve >= 0) { guard.armed = false; set_error("msm: this context has an enqueued MSM that was not finished"); return TRH_EBUSY; }
    return msm_host_tiled(bases->curve, scalars_host, nullptr, bases, offset, n, mont, out);
}

/* Params::commit / commit_lagrange for `batch` polynomials in HOST memory (one pointer per column, n scalars each): the columns
 * cross PCIe in chunks, chunk j + 1 while the batched MSM of chunk j runs. */
int trh_commit_batch_host(trh_bases_t bases, const uint64_t* const* polys_host, size_t n, size_t batch, const uint64_t* blinds_host, uint64_t* out) {
    TRH_TRY(msm_args(bases, 0, polys_host, n + 1, batch, out));
    if (!blinds_host) { set_error("commit_batch: null blinds"); return TRH_EINVAL; }
    if (bases->n != n + 1) { set_error("commit_batch: the handle must hold n + 1 = %zu bases (g or g_lagrange followed by w), it holds %zu", n + 1, bases->n); return TRH_EINVAL; }
    for (size_t i = 0; i < batch; ++i) if (!polys_host[i]) { set_error("commit_batch_host: column %zu is null", i); return TRH_EINVAL; }
    TRH_ENTER(0);
    Range range("trh_commit_batch_host");
    TRH_TRY(single_device(bases, "commit_batch_host"));
    Ctx& c = ctx();
    PendingGuard guard{c};
    if (c.msm.pending_curve >= 0) { guard.armed = false; set_error("commit_batch_host: this context has an enqueued MSM that was not finished"); return TRH_EBUSY; }
    TRH_TRY(stage_begin(c));
    StageScope scope(c);
    Stage& st = c.stage;
    size_t chunk = 16;  // 2^18-row columns: 128 MiB and ~7 ms of MSM per chunk
    while (chunk > 1 && chunk * n * 32 > ((size_t)256 << 20)) chunk >>= 1;
    if (chunk > batch) chunk = batch;
    const size_t nchunks = (batch + chunk - 1) / chunk;
    for (size_t j = 0; j < nchunks; ++j) {
        const size_t slot = j & 1, first = j * chunk, nb = first + chunk <= batch ? chunk : batch - first;
        TRH_TRY(st.ring_in[slot].ensure(chunk * n * 32 + chunk * 32 + 32));
        char* const tails = (char*)st.ring_in[slot].p + chunk * n * 32;
        for (size_t i = 0; i < nb; ++i) TRH_TRY(stage_h2d(c, (char*)st.ring_in[slot].p + i * n * 32, polys_host[first + i], n * 32, st.us, true));
        TRH_TRY(stage_h2d(c, tails, blinds_host + 4 * first, nb * 32, st.us));
        TRH_HIP_TRY(hipEventRecord(st.ev_up[slot], st.us));
        if (j > 0) TRH_TRY(msm_finish(bases->curve, st.cs, out + 12 * (first - chunk), chunk));
        TRH_HIP_TRY(hipStreamWaitEvent(st.cs, st.ev_up[slot], 0));
        TRH_TRY(msm_enqueue(bases->curve, bases->d_xy, lazy_bases(bases, 0, st.cs), st.ring_in[slot].p, n + 1, nb, n, 1, st.cs, fixed_base(bases, 0, n + 1), tails));
    }
    {
        const size_t first = (nchunks - 1) * chunk;
        TRH_TRY(msm_finish(bases->curve, st.cs, out + 12 * first, batch - first));
    }
    return scope.finish();
}

int trh_msm_dev_enqueue(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n, int mont, void* stream) {
    uint64_t dummy;
    TRH_TRY(msm_args(bases, offset, scalars_dev, n, 1, &dummy));
    TRH_ENTER(stream);
    Range range("trh_msm_dev_enqueue");
    TRH_TRY(single_device(bases, "msm_dev_enqueue"));
    if (ctx().msm.pending_curve >= 0) { set_error("msm_dev_enqueue: this context already has an MSM in flight (finish it, or use a second context: trh_ctx_create)"); return TRH_EBUSY; }
    TRH_TRY(msm_enqueue(bases->curve, (const char*)bases->d_xy + offset * 64, lazy_bases(bases, offset, (hipStream_t)stream), scalars_dev, n, 1, n, mont, (hipStream_t)stream, fixed_base(bases, offset, n)));
    ctx().msm.pending_owner = bases;
    return TRH_OK;
}
int trh_msm_dev_finish(trh_bases_t bases, void* stream, uint64_t out[12]) {
    if (!bases || !out) { set_error("msm_dev_finish: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_msm_dev_finish");
    if (ctx().msm.pending_curve < 0 || ctx().msm.pending_owner != (const void*)bases) { set_error("msm_dev_finish: no MSM over this base set is in flight on the calling context"); return TRH_EINVAL; }
    return msm_finish(bases->curve, (hipStream_t)stream, out, 1);
}
int trh_msm_dev(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n, int mont, void* stream, uint64_t out[12]) {
    TRH_TRY(msm_args(bases, offset, scalars_dev, n, 1, out));
    if (!bases->shards.empty()) return msm_sharded(bases, offset, scalars_dev, false, n, mont, (hipStream_t)stream, out);
    TRH_ENTER(stream);
    Range range("trh_msm_dev");
    TRH_TRY(single_device(bases, "msm_dev"));
    if (ctx().msm.pending_curve >= 0) { set_error("msm_dev: this context has an enqueued MSM that was not finished"); return TRH_EBUSY; }
    TRH_TRY(msm_enqueue(bases->curve, (const char*)bases->d_xy + offset * 64, lazy_bases(bases, offset, (hipStream_t)stream), scalars_dev, n, 1, n, mont, (hipStream_t)stream, fixed_base(bases, offset, n)));
    return msm_finish(bases->curve, (hipStream_t)stream, out, 1);
}
int trh_msm_batch_dev(trh_bases_t bases, size_t offset, const void* scalars_dev, size_t n, size_t batch, int mont, void* stream, uint64_t* out) {
    TRH_TRY(msm_args(bases, offset, scalars_dev, n, batch, out));
    TRH_ENTER(stream);
    Range range("trh_msm_batch_dev");
    TRH_TRY(single_device(bases, "msm_batch_dev"));
    if (ctx().msm.pending_curve >= 0) { set_error("msm_batch_dev: this context has an enqueued MSM that was not finished"); return TRH_EBUSY; }
    TRH_TRY(msm_enqueue(bases->curve, (const char*)bases->d_xy + offset * 64, lazy_bases(bases, offset, (hipStream_t)stream), scalars_dev, n, batch, n, mont, (hipStream_t)stream, fixed_base(bases, offset, n)));
    return msm_finish(bases->curve, (hipStream_t)stream, out, batch);
}
/* Params::commit / commit_lagrange for `batch` polynomials resident on the device: item b is the MSM of
 * polys[b] (n scalars) || blinds[b] over the n + 1 bases of the handle (g or g_lagrange followed by w) */
int trh_commit_batch_dev(trh_bases_t bases, const void* polys_dev, size_t n, size_t batch, const uint64_t* blinds_host, void* stream, uint64_t* out) {
    TRH_TRY(msm_args(bases, 0, polys_dev, n + 1, batch, out));
    if (!blinds_host) { set_error("commit_batch: null blinds"); return TRH_EINVAL; }
    if (bases->n != n + 1) { set_error("commit_batch: the handle must hold n + 1 = %zu bases (g or g_lagrange followed by w), it holds %zu", n + 1, bases->n); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_commit_batch_dev");
    TRH_TRY(single_device(bases, "commit_batch_dev"));
    Ctx& c = ctx();
    if (c.msm.pending_curve >= 0) { set_error("commit_batch_dev: this context has an enqueued MSM that was not finished"); return TRH_EBUSY; }
    hipStream_t s = (hipStream_t)stream;
    TRH_TRY(c.msm.tails.ensure(batch * 32));
    TRH_HIP_TRY(hipMemcpyAsync(c.msm.tails.p, blinds_host, batch * 32, hipMemcpyHostToDevice, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));  // the caller's blinds may be reused
    TRH_TRY(msm_enqueue(bases->curve, bases->d_xy, lazy_bases(bases, 0, s), polys_dev, n + 1, batch, n, 1, s, fixed_base(bases, 0, n + 1), c.msm.tails.p));
    return msm_finish(bases->curve, s, out, batch);
}

int trh_msm_set_window_bits(int cbits) {
    if (cbits != 0 && (cbits < 2 || cbits > 18)) { set_error("window bits must be 0 or in [2, 18]"); return TRH_EINVAL; }
    TRH_TRY(require_init());
    if (g_group.size() > 1 && thread_ctx() == g_default) {  // the group's shards follow the default context
        for (Ctx* g : g_group) { std::lock_guard<std::recursive_mutex> lk(g->mu); g->window_override = cbits; }
        return TRH_OK;
    }
    Ctx* c = thread_ctx();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->window_override = cbits;
    return TRH_OK;
}

int trh_point_sum(int curve, const uint64_t* pts, size_t count, uint64_t out[12]) {
    TRH_TRY(check_curve(curve));
    if (!out || (count && !pts)) { set_error("point_sum: null pointer"); return TRH_EINVAL; }
    return point_sum_host(curve, pts, count, out);
}

int trh_ntt_dev(int field, void* a_dev, uint32_t log_n, const uint64_t omega[4], size_t batch, void* stream) {
    TRH_TRY(check_field(field));
    if (!a_dev || !omega) { set_error("ntt_dev: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Range range("trh_ntt_dev");
    return ntt_device(field, a_dev, log_n, omega, batch, (hipStream_t)stream);
}

int trh_field_scale_rows_dev(int field, void* a_dev, size_t rows, size_t row_len, size_t active_len, const uint64_t* factors, uint32_t period, void* stream) {
    TRH_TRY(check_field(field));
    if (!a_dev || !factors || period == 0 || period > 64 || active_len > row_len) { set_error("field_scale: bad arguments"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    Ctx& c = ctx();
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
    TRH_TRY(check_field(field));
    TRH_ENTER(stream);
    if (!n) return TRH_OK;
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (field == TRH_FP) hipLaunchKernelGGL((field_op_kernel<FpParams>), dim3(gb), dim3(256), 0, (hipStream_t)stream, op, (const uint4*)a, (const uint4*)b, (uint4*)out, n);
    else hipLaunchKernelGGL((field_op_kernel<FqParams>), dim3(gb), dim3(256), 0, (hipStream_t)stream, op, (const uint4*)a, (const uint4*)b, (uint4*)out, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}
int trh_point_op_dev(int curve, int op, const void* p, const void* q, void* out, size_t n, void* stream) {
    TRH_TRY(check_curve(curve));
    TRH_ENTER(stream);
    if (!n) return TRH_OK;
    if (op == 3 || op == 4) {
        const unsigned gq = (unsigned)((n * 4 + 63) / 64);
        if (curve == TRH_PALLAS) hipLaunchKernelGGL((point_op_q4_kernel<FpParams>), dim3(gq), dim3(64), 0, (hipStream_t)stream, op, (const JacobianMem*)p, q, (JacobianMem*)out, n);
        else hipLaunchKernelGGL((point_op_q4_kernel<FqParams>), dim3(gq), dim3(64), 0, (hipStream_t)stream, op, (const JacobianMem*)p, q, (JacobianMem*)out, n);
        TRH_HIP_TRY(hipGetLastError());
        return TRH_OK;
    }
    const unsigned gb = (unsigned)((n + 63) / 64);
    if (curve == TRH_PALLAS) hipLaunchKernelGGL((point_op_kernel<FpParams>), dim3(gb), dim3(64), 0, (hipStream_t)stream, op, (const JacobianMem*)p, q, (JacobianMem*)out, n);
    else hipLaunchKernelGGL((point_op_kernel<FqParams>), dim3(gb), dim3(64), 0, (hipStream_t)stream, op, (const JacobianMem*)p, q, (JacobianMem*)out, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

// trh_malloc / trh_free keep freed blocks for reuse (per device, by rounded size): a host that allocates per call -- the polynomial
// side of multiopen makes two dozen short-lived n-element buffers per proof -- otherwise pays a map and an unmap of device memory
// (hundreds of microseconds each) around kernels of tens.  trh_free keeps hipFree's ordering: it returns once the device has
// finished everything queued, so a block never re-enters circulation under a kernel that still uses it.  TRH_POOL_MB caps the
// bytes kept (default 4096; 0: plain hipMalloc / hipFree); a failed allocation empties the pool and tries once more.
int trh_malloc(void** dev, size_t bytes) {
    if (!dev) { set_error("trh_malloc: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(0);
    if (!pool_cap()) {
        hipError_t e = hipMalloc(dev, bytes ? bytes : 16);
        if (e != hipSuccess) { set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return TRH_ENOMEM; }
        return TRH_OK;
    }
    const size_t rounded = pool_round(bytes);
    const int device = ctx().device;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (void* hit = g_pool.take(device, rounded)) {
        *dev = hit;
        return TRH_OK;
    } else {
        hipError_t e = hipMalloc(dev, rounded);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            pool_release_idle();
            e = hipMalloc(dev, rounded);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return TRH_ENOMEM; }
    }
    g_pool.live[*dev] = {device, rounded};
    return TRH_OK;
}
int trh_free(void* dev) {
    if (!dev) return TRH_OK;
    TRH_ENTER(0);
    std::unique_lock<std::mutex> lk(g_pool.mu);
    auto it = g_pool.live.find(dev);
    if (it == g_pool.live.end()) {  // not one of trh_malloc's (or the pool is off)
        if (g_pool.is_idle(dev)) { set_error("trh_free: block %p was already freed (it waits in the pool)", dev); return TRH_EINVAL; }
        lk.unlock();
        TRH_HIP_TRY(hipFree(dev));
        return TRH_OK;
    }
    const std::pair<int, size_t> key = it->second;
    g_pool.live.erase(it);
    lk.unlock();
    int prev = -1;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess && prev != key.first) e = hipSetDevice(key.first);
    if (e == hipSuccess) e = hipDeviceSynchronize();  // what hipFree would have waited for
    if (e != hipSuccess || key.second > pool_cap()) {
        // not poolable (or the device could not be drained): the block goes back to the runtime either way -- it has left `live`, so keeping
        // it would leak it and a second trh_free of the pointer would reach hipFree for a block the pool still believes it owns
        (void)hipGetLastError();
        const hipError_t ef = hipFree(dev);
        if (e == hipSuccess) e = ef;
    } else {
        lk.lock();
        // make room: drop idle blocks of THIS device, largest first (the keys sort by (device, size), so the device's largest block is
        // the last entry below (device + 1, 0)); only when the device has none left do other devices' blocks go
        while (g_pool.idle_bytes + key.second > pool_cap() && !g_pool.idle.empty()) {
            auto victim = g_pool.victim(key.first);
            int vprev = -1;
            (void)hipGetDevice(&vprev);
            (void)hipSetDevice(victim->first.first);
            (void)hipFree(victim->second);
            if (vprev >= 0) (void)hipSetDevice(vprev);
            g_pool.drop(victim);
        }
        g_pool.put_idle(dev, key);
        lk.unlock();
    }
    if (prev >= 0 && prev != key.first) (void)hipSetDevice(prev);
    TRH_HIP_TRY(e);
    return TRH_OK;
}
// gives the blocks trh_free kept back to the device (another allocator in the process -- torch's caching allocator, say -- cannot see them)
int trh_pool_trim(void) {
    pool_trim();
    return TRH_OK;
}
size_t trh_pool_idle_bytes(void) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    return g_pool.idle_bytes;
}
int trh_memcpy_h2d(void* dev, const void* host, size_t bytes) {
    TRH_ENTER(0);
    TRH_HIP_TRY(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return TRH_OK;
}
int trh_memcpy_d2h(void* host, const void* dev, size_t bytes) {
    TRH_ENTER(0);
    TRH_HIP_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return TRH_OK;
}
int trh_stream_synchronize(void* stream) {
    TRH_ENTER(stream);
    TRH_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return TRH_OK;
}

/* GPU-side timing for hosts without a HIP toolchain of their own (the Rust shim, examples/replay.cpp): events recorded on a stream between
 * the steps, read afterwards -- the time the device spent, without a host synchronisation (and the idle gap and clock ramp it causes)
 * between the steps */
int trh_event_create(void** out) {
    if (!out) { set_error("event_create: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(0);
    hipEvent_t e = nullptr;
    TRH_HIP_TRY(hipEventCreate(&e));
    *out = (void*)e;
    return TRH_OK;
}
int trh_event_record(void* event, void* stream) {
    if (!event) { set_error("event_record: null event"); return TRH_EINVAL; }
    TRH_ENTER(stream);
    TRH_HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return TRH_OK;
}
int trh_event_elapsed_ms(void* start, void* end, float* ms) {
    if (!start || !end || !ms) { set_error("event_elapsed_ms: null pointer"); return TRH_EINVAL; }
    TRH_TRY(require_init());
    TRH_HIP_TRY(hipEventSynchronize((hipEvent_t)end));
    TRH_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)end));
    return TRH_OK;
}
void trh_event_destroy(void* event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}

int trh_stat(const char* name, uint64_t* value) {
    if (!name || !value) { set_error("trh_stat: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(0);
    if (strcmp(name, "msm_lean_retries") == 0) { *value = ctx().msm.lean_retries; return TRH_OK; }
    if (strcmp(name, "msm_small_launches") == 0) { *value = ctx().msm.small_launches; return TRH_OK; }
    if (strcmp(name, "ipa_generator_collapses") == 0) { *value = ctx().ipa_collapses; return TRH_OK; }
    set_error("trh_stat: unknown counter '%s'", name);
    return TRH_EINVAL;
}

int trh_set_timing(int enabled) {
    TRH_TRY(require_init());
    Ctx* c = thread_ctx();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->timing = enabled;
    return TRH_OK;
}
int trh_last_timing(trh_timing_t* out) {
    if (!out) { set_error("last_timing: null pointer"); return TRH_EINVAL; }
    TRH_TRY(require_init());
    Ctx* c = thread_ctx();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    *out = c->last;
    return TRH_OK;
}

}  // extern "C"
