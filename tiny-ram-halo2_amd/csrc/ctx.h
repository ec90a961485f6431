// Device contexts of libtrh: error reporting, grow-only scratch buffers, timing.
//
// A context (Ctx) is bound to ONE device and owns every scratch buffer, table cache and pending state the entry points use,
// behind one lock.  trh_init() creates the process default; trh_ctx_create() makes further ones (a second lane on the same
// GPU so that two host threads overlap an MSM with an NTT, or one per GPU); trh_init_multi() makes the device group the
// range-sharded base sets run on.  Every extern "C" entry point opens with TRH_ENTER(stream): it resolves the calling thread's
// context, takes its lock, makes its device current for this thread (hipSetDevice is per thread: rayon workers never
// called trh_init) and orders the caller's stream behind the last stream that used the context's scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <array>
#include <functional>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/trh.h"
#include "curve.h"

struct trh_bases;
namespace trh { class CopyPool; }

namespace trh {

void set_error(const char* fmt, ...);

// The library's switches.  Parsed ONCE -- the environment variables TRH_<NAME> when the first option is needed, then
// whatever trh_set_option(name, value) overrides while no context exists -- into this struct, which nothing writes once a context is alive:
// no getenv on any call path (a host that runs libtrh from rayon workers while another thread calls setenv would race it).
struct Options {
    long pool_mb = 4096;     // TRH_POOL_MB        idle device blocks trh_malloc / trh_free keep per process (MiB)
    long stage_slot_mb = 16; // TRH_STAGE_SLOT_MB  pinned slot size of the host-pointer entries' rings (x 4 slots x 2 directions per context)
    int copy_threads = -1;   // TRH_COPY_THREADS   host threads per staging direction and context (-1: by core count)
    int bases_cache = 0;     // TRH_BASES_CACHE    1: trh_best_multiexp_* keeps base sets it has seen twice (full-content hash)
    int force_no_peer = 0;   // TRH_FORCE_NO_PEER  1: the device group hands device-resident scalars over through pinned host memory
    int trace = 0;           // TRH_TRACE          bit 0: timeline of the single-call host entries, bit 1: host side of the IPA's rounds (stderr)
    long msm_chunk_gb = 4;   // TRH_MSM_CHUNK_GB   scratch budget of one chunk of a batched MSM (GiB per digit array set)
    int sparse = 1;          // TRH_SPARSE         0: flag-like chunks of a fixed-base batch take the plain pipeline (no unit path)
    int reduce_q4 = 1;       // TRH_REDUCE_Q4      0: bucket reductions of small launches stay one thread per slice (no DPP-quad group law)
    int bin_sort = 1;        // TRH_BIN_SORT       0: the chunked bucket passes for every MSM (the path skewed scalars take anyway)
    int selftest = 1;        // TRH_SELFTEST       0: trh_init skips the known-answer self-test
    int ipa_fold = 1;        // TRH_IPA_FOLD       rounds after which the IPA opening collapses its generators (ipafold.hip); 0: never, 1: the library's choice
};
const Options& opt();

#define TRH_HIP_TRY(expr)                                                                    \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::trh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return TRH_EHIP;                                                                 \
        }                                                                                    \
    } while (0)

#define TRH_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc != TRH_OK) return _rc; \
    } while (0)

void pool_trim();  // capi.hip: returns the idle blocks of trh_malloc / trh_free to the device
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return TRH_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            pool_trim();  // the library's own scratch goes before blocks kept for the host's next allocation
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); p = nullptr; return TRH_ENOMEM; }
        cap = want;
        return TRH_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct TwiddleEntry {
    int field, log_n;
    u64 omega[4];
    DevBuf lo, hi;  // lo[i] = omega^i (i < 2^lo_bits), hi[i] = omega^(i << lo_bits)
    DevBuf zlo, zhi;  // the same in the lazy domain's Montgomery form (x 2^270, < 2 m)
    DevBuf direct[8]; // per pass p >= 1: omega^((k r) << shift) at [r * Ns + k], lazy form (absent when too large)
    DevBuf tile9;     // in-tile twiddles of a 9-stage pass, omega_512^i (i < 256), raw balanced limbs in three planes (signed passes)
    void release_all() { lo.release(); hi.release(); zlo.release(); zhi.release(); tile9.release(); for (DevBuf& d : direct) d.release(); }
    int lo_bits, hi_bits;
    u64 stamp;
    // a constant factor folded into the LAST pass's direct table (every element of that pass is multiplied by one of its entries anyway):
    // lagrange_to_coeff's 2^-k costs nothing this way.  Part of the cache key; scaled: the factor is in (the table exists)
    bool has_scale = false, scaled = false;
    u64 scale[4] = {0, 0, 0, 0};
};

// fixed-base table of an owned base set: table[j * n + i] = 2^(c j) * P_i (lazy affine form), j < W.  With it the
// digits of ALL windows go into one bucket set: one reduction per MSM instead of W, no Horner over windows.
struct MsmFixedBase {
    const void* table;
    int c, W;
};
struct MsmLane {         // scratch of one chunk of MSMs (leading dimension: batch item)
    DevBuf digits;       // W x n u32: bucket id | sign << 31
    DevBuf parted;       // W x n u32: entries grouped by level-1 bin (index | low bucket bits | sign)
    DevBuf sorted;       // W x n u32: point index | sign << 31, grouped by bucket
    DevBuf counts;       // W x nbins u32 level-1 histogram, then running cursor / bin end
    DevBuf bin_starts;   // W x nbins u32 level-1 bin start offsets
    DevBuf starts, ends; // W x (NB + 1) u32 bucket ranges in `sorted`
    DevBuf bucket_cnt;   // W x (NB + 1) u32 bucket histogram, then write cursors of the scatter pass
    DevBuf seg_bucket;   // W x nseg u32: bucket holding the first entry of each segment
    DevBuf first, last;  // W x nseg raw lazy XYZZ: first run / unfinished last run of each segment
    DevBuf direct;       // W x (NB + 1) raw lazy XYZZ: buckets that lie inside one segment
    DevBuf heavy;        // [0] count + list of (window, bucket) ids whose pieces a whole workgroup combines
    DevBuf buckets;      // W x NB XYZZ
    DevBuf partials;     // W x blocks XYZZ
    DevBuf sparse;       // sparse-column path: per item SP_LISTS list counters (one 128-byte line each), then one dense flag per item
};

struct MsmScratch {
    DevBuf scalars;      // host-scalar entry points stage here
    DevBuf tails;        // blinds of a commit batch (scalar n of every item)
    DevBuf bases_z;      // n affine bases converted to the lazy domain
    DevBuf window_sums;  // batch x W XYZZ
    MsmLane lane;
    bool dense_hint = false;    // the caller knows its scalars are full-size (the IPA's round MSMs): the sparse classifier is skipped, the combine takes the quad form
    // lean sort (msm.hip): the enqueued MSM skipped the chunked fallback passes; msm_finish checks the flags and repeats it with them if needed
    bool force_fallback = false, lean_pending = false;
    struct { const void *bases_dev, *bases_z, *scalars_dev, *tails_dev; size_t n, batch, stride; int mont; bool has_fb; MsmFixedBase fb; } retry{};
    unsigned lean_retries = 0;  // how often that happened on this context (tests)
    u64 small_launches = 0;     // MSMs that ran as one msm_small_kernel launch (tests: the path was taken, not fallen back from)
    size_t lean_off_n = 0; int lean_off_c = 0;  // the shape (pairs, window bits) whose lean sort last overflowed: the fallback launches are queued for it again
    bool reserve_only = false;  // msm_enqueue sizes the scratch of the described launch and returns before the first kernel (trh_bases_reserve)
    bool no_sparse_vote = false;  // the sparse classifier is skipped and nothing else changes (the shards of a range-sharded MSM: its host synchronisation would hold back the other shards)
    void* sp_host = nullptr;    // pinned: the sparse path's list counters as read back, then the dense flags it sends down
    void* host_sums = nullptr;  // pinned mirror of window_sums
    size_t host_sums_cap = 0;
    // state of the enqueued-but-not-finished MSM
    int pending_curve = -1, pending_windows = 0, pending_c = 0;
    size_t pending_batch = 0;
    hipStream_t pending_stream = nullptr;  // the finish must name the stream (and, through the C ABI, the base set) of its enqueue
    const void* pending_owner = nullptr;
    // an MSM beyond MSM_TILE pairs runs as range tiles: the sum of the finished tiles (normalised Jacobian), added by msm_finish
    bool tile_sum_valid = false, in_tile = false;
    u64 tile_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
};

// Host <-> device staging of the host-pointer entry points (hostio.hip): pinned slot rings + three streams.  Measured on the
// MI355X box (profiles/pcie_probe_r03.txt): the runtime's own pageable path pins the caller's pages at first sight -- 27 GB/s up,
// 12.7 GB/s down for a buffer it has not seen, hipMemcpyAsync blocks -- while pinned copies run at 57 GB/s and both directions at
// once at 2 x 48 GB/s.  The rings carry a pageable buffer in slots: a pool of host threads copies slot i + 1 while the DMA engine
// moves slot i, uploads, kernels and downloads of a batch run on their own streams.
struct Stage {
    static constexpr int NS = 4;
    size_t slot = 0;                 // bytes per slot
    char* up = nullptr;              // NS slots of pinned memory, host -> device
    char* down = nullptr;            // NS slots, device -> host
    hipEvent_t up_ev[NS] = {}, down_ev[NS] = {};
    bool up_used[NS] = {};
    unsigned up_next = 0;
    hipStream_t us = nullptr, cs = nullptr, ds = nullptr;  // upload, compute, download
    hipEvent_t ev_up[4] = {nullptr, nullptr, nullptr, nullptr}, ev_comp[4] = {nullptr, nullptr, nullptr, nullptr};
    DevBuf ring_in[4], ring_out[4];  // device-side ring of the batch pipelines
    // traffic of the host-pointer entry points on this context since trh_io_stats_reset (bytes, host seconds inside the copies)
    double up_bytes = 0, down_bytes = 0, up_s = 0, down_s = 0;
    double up_zero_bytes = 0;        // of up_bytes: slots that were zero throughout and became a device-side memset instead of a DMA
    // this context's copy threads (hostio.hip; created at first use, joined in stage_release): one pool per direction, so that uploads and a
    // pipeline's download helper never queue behind each other, and per CONTEXT, so that the GPUs of a device group are fed in parallel
    CopyPool* up_pool = nullptr;
    CopyPool* down_pool = nullptr;
    std::map<int, std::array<hipEvent_t, NS>> xfer_ev;  // stage_d2d_via_host: "slot filled" events on the source device, per source device
};

struct Ctx {
    std::recursive_mutex mu;  // recursive: an entry point may call another one (and a transcript callback may call host-side helpers)
    bool inited = false;
    int device = -1;
    int timing = 0;
    int window_override = 0;
    trh_timing_t last{};
    MsmScratch msm;
    DevBuf ntt_tmp;
    DevBuf io;  // staging for host-pointer NTT entry points
    Stage stage;
    DevBuf pfft;  // curve-point FFT work array + twiddle scalars
    DevBuf scan, scan2;  // prefix-product block totals / batch-inversion running products
    DevBuf ipa[11];  // vectors of the IPA prover (b, s', p', weights, round scalars, g‖w‖u and its lazy copy, the second halves of the p' / b ping-pong pairs,
                     // the generator fold's buckets and bucket lists -- ipafold.hip), kept across proofs
    DevBuf factors;  // ring of 16 small factor tables for the scale kernels
    unsigned factor_slot = 0;
    void* pinned_ring = nullptr;  // 64 x 2 KiB of pinned host memory mirroring the factor ring: constants are copied here first, so the
    void* pinned_land = nullptr;  // 4 KiB pinned landing area for the few words the host reads back per IPA round (a pageable target costs a staging copy)
    unsigned pinned_slot = 0;     // asynchronous upload never reads a caller's stack buffer and needs no synchronisation
    void* pinned_fold = nullptr;  // pinned source of the generator fold's bucket lists (ipafold.hip)
    size_t pinned_fold_cap = 0;
    u64 ipa_collapses = 0;      // openings that collapsed their generators (ipafold.hip; tests)
    bool helper_failed = false;
    class HostHelper* helper = nullptr;  // host thread for the second half of a batch's Horners (hosthelper.h; msm_finish)
    std::vector<TwiddleEntry*> twiddles;
    u64 stamp = 0;
    void* lookup_scratch = nullptr;  // lookup.hip's buffers (opaque here)
    int cu_count = 0;                // compute units of the device (persistent kernels size their grids by it); 0: not asked yet
    unsigned attr_done = 0;          // hipFuncSetAttribute is per device: bit per kernel family already configured for this context's device
    hipStream_t own_stream = nullptr;  // the multi-device MSM driver enqueues this context's shard here
    // scratch is shared by every stream that enters this context: a call on another stream than the previous one waits for it
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t order_ev = nullptr;
};
enum { ATTR_MSM = 1u, ATTR_NTT = 2u, ATTR_EXPR = 4u };

Ctx& ctx();  // the context entered (TRH_ENTER) by the calling thread
int require_init();

// Scope of one entry point: resolve the thread's context (trh_ctx_set_current, else the process default of trh_init), lock it,
// make its device current for the calling thread, order `stream` behind the context's previous stream.
struct Enter {
    Ctx* c = nullptr;
    Ctx* prev_active = nullptr;
    int prev_device = -1;
    bool locked = false, outermost = false;
    hipStream_t entered_stream = nullptr;
    int begin(hipStream_t stream, Ctx* explicit_ctx = nullptr);
    ~Enter();
};
#define TRH_ENTER(stream)  \
    ::trh::Enter _trh_enter; \
    TRH_TRY(_trh_enter.begin((hipStream_t)(stream)))
#define TRH_ENTER_CTX(stream, cptr)  \
    ::trh::Enter _trh_enter; \
    TRH_TRY(_trh_enter.begin((hipStream_t)(stream), (cptr)))

// roctx range around a primitive (rocprofv3 --marker-trace); resolved with dlopen, a no-op when the library is absent
struct Range {
    explicit Range(const char* name);
    ~Range();
};

// hostio.hip
int stage_ensure(Ctx& c);
void stage_release(Ctx& c);
// src_host -> dst_dev on stream s through the upload ring; returns when the source has been read (the last DMA may still be in flight on s)
// (part_of_batch: more uploads follow at once -- full slots throughout, the next call's copy overlaps this one's DMA; otherwise the
//  transfer starts and ends with short chunks; head_only: short chunks at the start only -- the first of a run of uploads)
// (zero_elide: slots whose source is zero throughout become a hipMemsetAsync -- the zero-padded vectors of coeff_to_extended)
// (speculate: chunks whose PROBE -- one 64-byte line per 64 KiB, the first and the last -- is zero are cleared on the device at once and
//  listed as (offset, length) instead of being read through; the caller must verify every listed range before it relies on the result)
int stage_h2d(Ctx& c, void* dst_dev, const void* src_host, size_t bytes, hipStream_t s, bool part_of_batch = false, bool zero_elide = false, bool head_only = false,
              std::vector<std::pair<size_t, size_t>>* speculate = nullptr);
// src (device src_device, ordered behind src_stream) -> dst on dstc's device WITHOUT peer access: slots of dstc's pinned download ring carry
// the bytes (D2H on src_stream, H2D on dst_stream), everything stream-ordered, no host synchronisation
int stage_d2d_via_host(Ctx& dstc, void* dst_dev, hipStream_t dst_stream, const void* src_dev, int src_device, hipStream_t src_stream, size_t bytes);
// src_dev -> dst_host through the download ring, ordered behind the work queued on s; returns when dst_host is complete.
// before_copy_out (optional) runs once after the first DMAs into the ring were issued and before anything is written to dst_host; a
// non-zero return stops the download with that code (dst_host untouched, the DMAs already issued only touch the ring)
int stage_d2h(Ctx& c, void* dst_host, const void* src_dev, size_t bytes, hipStream_t s, const std::function<int()>* before_copy_out = nullptr);
// Batch pipeline over `count` items (each a group of host buffers): upload (caller thread, stage.us) -> compute(item, in, out,
// stage.cs) -> download (helper thread, stage.ds), over a ring of device buffers.  in_bytes / out_bytes: device bytes per item;
// upload(item, dev_in) issues the stage_h2d calls of one item; segments(item, dev_out, list) names where its results go -- the helper
// thread keeps the download ring full ACROSS items (the DMA of item i + 1 starts while the last slots of item i are copied out).
struct HostPipe {
    size_t count = 0, in_bytes = 0, out_bytes = 0;
    bool in_place = false;  // compute works in the input buffer (out == in)
    std::function<int(size_t, void*)> upload;
    std::function<int(size_t, void*, void*, hipStream_t)> compute;
    // the host destinations of one item: (dst_host, src_dev, bytes) segments, src inside the item's device output buffer
    struct Seg { void* dst; const void* src; size_t bytes; };
    std::function<void(size_t, const void*, std::vector<Seg>&)> segments;
};
int host_pipeline(Ctx& c, const HostPipe& p);
int stage_begin(Ctx& c);  // the stage's streams wait for the context's previous work
int stage_end(Ctx& c);    // drains the three streams
// Scope of a single-call host entry between stage_begin and its return: on EVERY path out (an error return through TRH_TRY included)
// the three stage streams are drained first -- kernels and DMAs queued there may still be touching c.io, the rings and the caller's
// page-locked slice, and the next entry only orders itself behind the stream the context was entered with.
struct StageScope {
    Ctx& c;
    bool done = false;
    explicit StageScope(Ctx& c_) : c(c_) {}
    int finish() { done = true; return stage_end(c); }
    ~StageScope() {
        if (done) return;
        (void)hipStreamSynchronize(c.stage.us); (void)hipStreamSynchronize(c.stage.cs); (void)hipStreamSynchronize(c.stage.ds);
    }
};
int best_fft_host(int field, uint64_t* a, const uint64_t* omega, uint32_t log_n);
// ntt.hip
// pointwise steps of EvaluationDomain fused into the first / last pass of a transform (lazy passes only:
// ntt_can_fuse).  Factor tables hold `period` elements in the lazy Montgomery form (x 2^270, < 2 m), device memory.
struct NttFusion {
    const void* in_dev = nullptr;  // pass 0 reads rows of 2^in_log elements from here, zero beyond (zero-padding); null: a_dev
    uint32_t in_log = 0;
    const void* pre = nullptr;     // x[i] *= pre[i % pre_period] on load (i = index inside the row)
    uint32_t pre_period = 0;
    const void* post = nullptr;    // y[i] *= post[i % post_period] on the final store
    uint32_t post_period = 0;
    // coset blocks (EvaluationDomain's extended domain as 2^(extended_k - k) cosets of size 2^k, domain.hip): transform t of the batch
    // belongs to block t % blocks and -- with pre_blocks -- reads input row t / blocks (in_dev, 2^in_log = 2^log_n elements per row).
    // Tables: [blocks][2^log_n] entries as raw balanced limbs in three planes (ntt_block_table layout, 36 B per entry).
    const void* pre_blocks = nullptr;   // x[i] *= pre_blocks[t % blocks][i] on the loads of pass 0
    const void* post_blocks = nullptr;  // y[i] *= post_blocks[t % blocks][i] on the final store
    uint32_t blocks = 0;
};
// bytes of a [blocks][2^log_n] table of NttFusion::pre_blocks / post_blocks, and the kernel that fills one: entry (b, i) =
// base_b^i * scale with base_b = g * w^b (canonical Montgomery inputs): the coset generators' powers
size_t ntt_block_table_bytes(uint32_t blocks, uint32_t log_n);
int ntt_block_table_build(int field, void* table_dev, uint32_t blocks, uint32_t log_n, const u64 g[4], const u64 w[4], const u64 scale[4], hipStream_t s);
// out[t][i] = in[t / blocks][i] * table[t % blocks][i] (canonical words in and out): the unfused form for sizes below the lazy passes
int ntt_block_scale(int field, const void* in_dev, void* out_dev, size_t transforms, uint32_t blocks, uint32_t log_n, const void* table_dev, bool in_per_block, hipStream_t s);
bool ntt_can_fuse(uint32_t log_n);
int ntt_lazy_shift();
int ntt_device(int field, void* a_dev, uint32_t log_n, const u64 omega[4], size_t batch, hipStream_t s, const NttFusion* fu = nullptr, const u64* scale = nullptr);
// the tables of that transform (built if this context does not hold them yet) and the scratch for `batch` transforms at a time, without running one
int ntt_prepare(int field, uint32_t log_n, const u64 omega[4], const u64* scale, size_t batch, uint32_t blocks, hipStream_t s);
// whether a transform of this size can take a constant factor (Montgomery words) in its last pass's table: ntt_device(..., scale) then returns a . scale
bool ntt_can_fold_scale(uint32_t log_n);
void ntt_release_tables();
// ipafold.hip: the IPA's generators after r collapses, from the fixed-base table of g || w || u
bool ipa_fold_supported(const MsmFixedBase& fb, uint32_t k, uint32_t r);
int ipa_fold_reserve(const MsmFixedBase& fb, uint32_t k, uint32_t r);
int ipa_reserve(int curve, const trh_bases* gw, uint32_t k);  // ipa.hip: the opening's vectors and the collapse's buffers, at setup time
int ipa_fold_generators(int curve, const MsmFixedBase& fb, size_t row, uint32_t k, uint32_t r, const u64* u_mont, void* out_xy, void* out_z, hipStream_t s);
// msm.hip
int msm_enqueue(int curve, const void* bases_dev, const void* bases_z_or_null, const void* scalars_dev, size_t n, size_t batch,
                size_t scalar_stride_elems, int mont, hipStream_t s, const MsmFixedBase* fb = nullptr, const void* tails_dev = nullptr);
size_t msm_small_max_pairs();  // the largest MSM msm_small_kernel takes (batches of up to four, no window override)
int msm_fixed_base_windows(int c);
bool msm_fixed_base_fits(size_t n, int c);
int msm_build_table(int curve, const void* bases_dev, size_t n, int c, void* table_dev, hipStream_t s);
// One base in the form the accumulation reads: 128 bytes = limbs 0 .. 7 of x, of y and of -y (eight words each, 29-bit limbs of the
// signed domain of field.h), then the three top limbs and five spare words.  The mixed addition takes the limbs as they are -- no
// unpacking of 32-bit words into 29-bit limbs, no negation for a negative digit (it reads -y instead of y); both 64-byte halves of
// the record are read either way, which doubles the gather's bytes and removes 50 of the ~1900 instructions of an addition.
constexpr size_t ZREC = 128;
int msm_convert_bases(int curve, const void* in_dev, void* out_dev, size_t n, hipStream_t s);
int msm_finish(int curve, hipStream_t s, u64* out_xyz, size_t batch);
int point_sum_host(int curve, const u64* pts, size_t count, u64* out);
int bases_generate_device(int curve, u64 s0, u64 d, u64 first, size_t n, void* out_dev, hipStream_t s);
void msm_release();
void lookup_release();

}  // namespace trh

struct trh_bases {
    int curve;
    void* d_xy;
    size_t n;
    bool owned;
    void* d_z = nullptr;  // owned (immutable) sets: the bases converted once to the lazy Montgomery domain
    void* d_table = nullptr;  // trh_bases_precompute: W x n shifted copies (see MsmFixedBase)
    trh::MsmFixedBase fb{nullptr, 0, 0};
    trh::Ctx* owner = nullptr;  // the context (device) the memory lives on
    // handles are shared by every context of their device: the lazily built copies (d_z, d_table) are created / replaced under this lock
    // (two threads on two contexts running their first MSM over the same set would otherwise both convert, and one copy would leak)
    std::mutex mu;
    // range-sharded set (trh_init_multi): shard g holds bases [shard_off[g], shard_off[g + 1]) on its own context's device;
    // d_xy is null and the per-shard handles carry the device memory
    std::vector<trh_bases*> shards;
    std::vector<size_t> shard_off;
};
