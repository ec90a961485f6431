// Radix-2 number-theoretic transform over Fp / Fq for gfx950.
//
// Replaces halo2_proofs 0.2.0 `arithmetic::best_fft` (arithmetic.rs; crate pinned at
// /root/reference/Cargo.lock:619-621) as called by `poly::EvaluationDomain::{lagrange_to_coeff,
// coeff_to_extended, extended_to_coeff}` inside keygen_* / create_proof
// (/root/reference/src/test_utils.rs:23-25, 41-49).  Same contract: in place, natural order in
// and out, a'[i] = sum_j a[j] omega^(i j), omega a primitive 2^log_n-th root of unity.
//
// Schedule (Stockham auto-sort, so no bit-reversal pass over HBM): log_n is split into passes of
// s <= 9 stages.  One workgroup owns a tile of R = 2^s rows x C columns (R*C = 2048 elements =
// 64 KiB of LDS): it reads C-element runs at stride N/R, applies the inter-pass twiddle
// omega^(k r), runs the s radix-2 stages entirely in LDS with an LDS-resident table of the R/2
// in-tile twiddles, and writes C-element runs to the auto-sorted position.  Passes ping-pong
// between the buffer and a scratch buffer; the last pass of an odd count is in place (its tile
// reads and writes the same addresses).  Twiddles omega^e come from two L2-resident tables
// (omega^lo, omega^(hi << lo_bits)) built on the device per (field, log_n, omega).
//
// Integer work, no MFMA.  Algorithmic HBM bytes: 32 B read + 32 B write per element.
#include <stdlib.h>
#include <string.h>

#include "ctx.h"

namespace trh {

namespace {

constexpr int TILE_LOG = 11;
constexpr int TILE = 1 << TILE_LOG;
constexpr int NTT_THREADS = TILE / 4;
constexpr int MAX_PASS_LOG = 9;

struct alignas(16) Half {  // 16 bytes of a field element
    u32 w[4];
};

template <class F>
__device__ __forceinline__ Fe<F> load_fe(const uint4* __restrict__ p) {
    uint4 a = p[0], b = p[1];
    return fe_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
template <class F>
__device__ __forceinline__ void store_fe(uint4* __restrict__ p, const Fe<F>& v) {
    u32 w[8];
    fe_store(v, w);
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// LDS holds each element in memory format as two 16-byte halves in separate planes: consecutive
// lanes touch consecutive 16-byte slots, which ds_read_b128 / ds_write_b128 serve conflict-free
template <class F>
__device__ __forceinline__ Fe<F> lds_load(const uint4* lo, const uint4* hi, int idx) {
    uint4 a = lo[idx], b = hi[idx];
    return fe_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
template <class F>
__device__ __forceinline__ void lds_store(uint4* lo, uint4* hi, int idx, const Fe<F>& v) {
    u32 w[8];
    fe_store(v, w);
    lo[idx] = make_uint4(w[0], w[1], w[2], w[3]);
    hi[idx] = make_uint4(w[4], w[5], w[6], w[7]);
}

// omega^e from the two-level tables
template <class F>
__device__ __forceinline__ Fe<F> twiddle(const uint4* __restrict__ t_lo, const uint4* __restrict__ t_hi, u32 e, int lo_bits) {
    const u32 el = e & ((1u << lo_bits) - 1u), eh = e >> lo_bits;
    Fe<F> w = load_fe<F>(t_lo + 2 * (size_t)el);
    if (eh) w = fe_mul(w, load_fe<F>(t_hi + 2 * (size_t)eh));
    return w;
}

// tables: lo[i] = omega^i, hi[i] = omega^(i << lo_bits); pw[b] = omega^(2^b) supplied by the host
template <class F>
__global__ void __launch_bounds__(256) ntt_tables_kernel(const uint4* __restrict__ pw, uint4* __restrict__ t_lo, uint4* __restrict__ t_hi, int lo_bits, int hi_bits) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nlo = 1u << lo_bits, nhi = 1u << hi_bits;
    if (i < nlo) {
        Fe<F> r = fe_one<F>();
        for (int b = 0; b < lo_bits; ++b)
            if ((i >> b) & 1u) r = fe_mul(r, load_fe<F>(pw + 2 * b));
        store_fe<F>(t_lo + 2 * (size_t)i, r);
    }
    if (i < nhi) {
        Fe<F> r = fe_one<F>();
        for (int b = 0; b < hi_bits; ++b)
            if ((i >> b) & 1u) r = fe_mul(r, load_fe<F>(pw + 2 * (b + lo_bits)));
        store_fe<F>(t_hi + 2 * (size_t)i, r);
    }
}
// the same tables in the lazy domain's Montgomery form (x 2^270 unsigned / x 2^261 signed), values < 2 m
template <class F, bool SIGNED>
__global__ void __launch_bounds__(256) ntt_tables_lazy_kernel(const uint4* __restrict__ t, uint4* __restrict__ z, u32 cnt) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    u32 w[8];
    if (SIGNED) fy_store(fy_from_fe(load_fe<F>(t + 2 * (size_t)i)), w);  // non-negative, below m (1 + 2^-7)
    else fz_store(fz_from_fe(load_fe<F>(t + 2 * (size_t)i)), w);
    z[2 * (size_t)i] = make_uint4(w[0], w[1], w[2], w[3]);
    z[2 * (size_t)i + 1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// One Stockham pass: R = 2^s, Ns = 2^log_ns (size of the sub-transforms already done).
template <class F>
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int log_n, int s, int log_ns,
                                                               const uint4* __restrict__ t_lo, const uint4* __restrict__ t_hi, int lo_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = 1 << s;
    const int log_c = (log_n < TILE_LOG ? log_n : TILE_LOG) - s;  // columns per tile
    const int C = 1 << log_c;
    const int E = R << log_c;  // elements in this tile
    uint4* lds_lo = (uint4*)smem;
    uint4* lds_hi = lds_lo + E;
    uint4* tw_lo = lds_hi + E;        // R/2 in-tile twiddles omega_R^i, same split layout
    uint4* tw_hi = tw_lo + (R >> 1);

    const size_t N = (size_t)1 << log_n;
    const size_t batch_off = (size_t)blockIdx.y * N * 2;  // in uint4 units
    in += batch_off;
    out += batch_off;
    const u32 j0 = blockIdx.x << log_c;
    const u32 ns_mask = (1u << log_ns) - 1u;
    const int tid = threadIdx.x;
    const size_t row_stride = N >> s;  // N / R

    // in-tile twiddle table: omega_R^i = omega^(i << (log_n - s))
    for (int i = tid; i < (R >> 1); i += NTT_THREADS) {
        Fe<F> w = twiddle<F>(t_lo, t_hi, (u32)i << (log_n - s), lo_bits);
        lds_store<F>(tw_lo, tw_hi, i, w);
    }
    // load + inter-pass twiddle + bit-reversed row placement
    const int tw_shift = log_n - log_ns - s;  // exponent scale N / (Ns R)
    for (int e = tid; e < E; e += NTT_THREADS) {
        const u32 c = e & (C - 1), r = e >> log_c;
        const u32 j = j0 + c, k = j & ns_mask;
        Fe<F> x = load_fe<F>(in + 2 * ((size_t)j + (size_t)r * row_stride));
        if (log_ns > 0) {
            const u32 ex = (k * r) << tw_shift;
            if (ex) x = fe_mul(x, twiddle<F>(t_lo, t_hi, ex, lo_bits));
        }
        const u32 rr = __brev(r) >> (32 - s);
        lds_store<F>(lds_lo, lds_hi, (int)((rr << log_c) | c), x);
    }
    __syncthreads();
    // s radix-2 DIT stages in LDS
    for (int st = 0; st < s; ++st) {
        const int half = 1 << st;
        for (int bf = tid; bf < (E >> 1); bf += NTT_THREADS) {
            const int c = bf & (C - 1), p = bf >> log_c;
            const int pos = p & (half - 1);
            const int r0 = ((p >> st) << (st + 1)) | pos;
            const int i0 = (r0 << log_c) | c, i1 = i0 + (half << log_c);
            Fe<F> a = lds_load<F>(lds_lo, lds_hi, i0);
            Fe<F> b = lds_load<F>(lds_lo, lds_hi, i1);
            if (pos) b = fe_mul(b, lds_load<F>(tw_lo, tw_hi, pos << (s - 1 - st)));
            lds_store<F>(lds_lo, lds_hi, i0, fe_add(a, b));
            lds_store<F>(lds_lo, lds_hi, i1, fe_sub(a, b));
        }
        __syncthreads();
    }
    // store to the auto-sorted position
    for (int e = tid; e < E; e += NTT_THREADS) {
        const u32 c = e & (C - 1), r = e >> log_c;
        const u32 j = j0 + c, k = j & ns_mask;
        const size_t dst = ((size_t)(j - k) << s) + k + ((size_t)r << log_ns);
        store_fe<F>(out + 2 * dst, lds_load<F>(lds_lo, lds_hi, e));
    }
}

// DIT butterfly: (a, b) -> (a + b, a - b)
template <class F>
__device__ __forceinline__ void bfly(Fe<F>& a, Fe<F>& b) {
    const Fe<F> t = fe_add(a, b);
    b = fe_sub(a, b);
    a = t;
}

// Same Stockham pass for full 2048-element tiles, with the stages grouped LG at a time in
// registers: a thread owns G = 2^LG rows of one column (LG = 2: 512 threads, 4 rows; 8 rows spill).  Round 0 takes
// its rows straight from HBM (rows m + v R/G, i.e. the bit-reversed neighbours rr..rr+G-1), applies the
// inter-pass twiddle and runs stages 0..LG-1, whose twiddles are the constants 1, w4, w8, w8^3; later
// rounds exchange through LDS (one read + one write per element per LG stages, in place, one barrier per
// round); the last round writes its results straight to HBM.
template <class F, int LG, int TLOG>
__global__ void __launch_bounds__((1 << TLOG) >> LG) ntt_passg_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int log_n, int s, int log_ns,
                                                               const uint4* __restrict__ t_lo, const uint4* __restrict__ t_hi, int lo_bits) {
    constexpr int G = 1 << LG, T = 1 << TLOG, THREADS = T >> LG;  // T elements per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = 1 << s;
    const int log_c = TLOG - s;
    const int C = 1 << log_c;
    uint4* lds_lo = (uint4*)smem;
    uint4* lds_hi = lds_lo + T;
    uint4* tw_lo = lds_hi + T;
    uint4* tw_hi = tw_lo + (R >> 1);

    const size_t N = (size_t)1 << log_n;
    const size_t batch_off = (size_t)blockIdx.y * N * 2;
    in += batch_off;
    out += batch_off;
    const int tid = threadIdx.x;
    const u32 c = tid & (C - 1), m = tid >> log_c;  // m in [0, R/G)
    const u32 j = (blockIdx.x << log_c) + c;
    const u32 k = j & ((1u << log_ns) - 1u);
    const size_t row_stride = N >> s;

    for (int i = tid; i < (R >> 1); i += THREADS) lds_store<F>(tw_lo, tw_hi, i, twiddle<F>(t_lo, t_hi, (u32)i << (log_n - s), lo_bits));

    Fe<F> x[G];
    const int tw_shift = log_n - log_ns - s;
#pragma unroll
    for (int v = 0; v < G; ++v) {
        const u32 r = m + (u32)v * (u32)(R >> LG);
        Fe<F> val = load_fe<F>(in + 2 * ((size_t)j + (size_t)r * row_stride));
        if (log_ns > 0) {
            const u32 ex = (k * r) << tw_shift;
            if (ex) val = fe_mul(val, twiddle<F>(t_lo, t_hi, ex, lo_bits));
        }
        x[(int)(__builtin_bitreverse32((u32)v) >> (32 - LG))] = val;  // element v sits at bit-reversed slot u (v is a compile-time constant)
    }
    __syncthreads();  // in-tile twiddle table complete

    u32 base = (s > LG) ? ((__brev(m) >> (32 - (s - LG))) << LG) : 0u;  // rows rr = base + (u << stl)
    u32 L = 0;
    int stl = 0, vb = 0;
    for (int st = 0; st < s; st += LG) {
        if (st > 0) {
            // hand the finished rows over through LDS and pick up the next group of G
#pragma unroll
            for (int u = 0; u < G; ++u) lds_store<F>(lds_lo, lds_hi, (int)(((base + ((u32)u << stl)) << log_c) | c), x[u]);
            __syncthreads();
            stl = st + LG <= s ? st : s - LG;
            vb = st - stl;  // stages below vb of this layout were done in the previous round
            L = m & ((1u << stl) - 1u);
            base = L | ((m >> stl) << (stl + LG));
#pragma unroll
            for (int u = 0; u < G; ++u) x[u] = lds_load<F>(lds_lo, lds_hi, (int)(((base + ((u32)u << stl)) << log_c) | c));
        }
#pragma unroll
        for (int v = 0; v < LG; ++v) {
            if (v >= vb) {
                const int sh = s - 1 - stl - v;
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    if (u & (1 << v)) continue;
                    const u32 idx = (L + ((u32)(u & ((1 << v) - 1)) << stl)) << sh;
                    if (idx) x[u | (1 << v)] = fe_mul(x[u | (1 << v)], lds_load<F>(tw_lo, tw_hi, (int)idx));
                    bfly(x[u], x[u | (1 << v)]);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const u32 rr = base + ((u32)u << stl);
        const size_t dst = ((size_t)(j - k) << s) + k + ((size_t)rr << log_ns);
        store_fe<F>(out + 2 * dst, x[u]);
    }
}

// ---- lazy-domain pass (the default for full tiles) -------------------------------------------
// Same schedule as ntt_passg_kernel, but the values stay unreduced nine-limb residues (Fz) for the
// whole pass: the input words x R (canonical Montgomery, or < 2 m from a previous pass) are taken as
// they are, every twiddle is held in the 2^270 Montgomery form so that fz_mul(v, W) = v w carries the
// element's own 2^256 factor through, and a butterfly is one fz_mul + one carry chain each for
// a + t and a + K m - t: no conditional subtraction, no 16-bit round, and the LDS exchange moves the
// nine limbs as they are (two 16-byte planes + one 4-byte plane: no pack / unpack).
// Bounds (in units of m, inputs < 2): round 0 does stages 0..LG-1 with the trivial twiddle left out,
// so stage v sees operands < 2^(v+1) and leaves < 2^(v+2); every later stage multiplies (t < 2) and
// adds 2.  After s <= 9 stages the values are < 2^(LG+1) + 2 (s - LG) <= 30 (LG = 3: 28), far below the 256 m
// that fz_mul tolerates against a twiddle < 2 m, and nine limbs hold 2^16 m.  The pass ends with
// v - max(floor(v / 2^254) - 1, 0) m in [0, 2 m) and, on the last pass, the conditional subtraction.
template <class F>
__device__ __forceinline__ Fz<F> load_fz(const uint4* __restrict__ p) {
    uint4 a = p[0], b = p[1];
    return fz_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
template <class F>
__device__ __forceinline__ Fz<F> lds_load_words(const uint4* lo, const uint4* hi, int idx) {
    uint4 a = lo[idx], b = hi[idx];
    return fz_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
template <class F>
__device__ __forceinline__ void lds_store_words(uint4* lo, uint4* hi, int idx, const Fz<F>& v) {
    u32 w[8];
    fz_store(v, w);
    lo[idx] = make_uint4(w[0], w[1], w[2], w[3]);
    hi[idx] = make_uint4(w[4], w[5], w[6], w[7]);
}
template <class F>
__device__ __forceinline__ Fz<F> lds_load_limbs(const uint4* pa, const uint4* pb, const u32* pc, int idx) {
    const uint4 a = pa[idx], b = pb[idx];
    Fz<F> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = pc[idx];
    return r;
}
template <class F>
__device__ __forceinline__ void lds_store_limbs(uint4* pa, uint4* pb, u32* pc, int idx, const Fz<F>& v) {
    pa[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    pb[idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    pc[idx] = v.l[8];
}
template <class F>
__device__ __forceinline__ Fz<F> twiddle_z(const uint4* __restrict__ z_lo, const uint4* __restrict__ z_hi, u32 e, int lo_bits) {
    const u32 el = e & ((1u << lo_bits) - 1u), eh = e >> lo_bits;
    Fz<F> w = load_fz<F>(z_lo + 2 * (size_t)el);
    if (eh) w = fz_mul(w, load_fz<F>(z_hi + 2 * (size_t)eh));
    return w;
}
// (a, b) -> (a + b, a + K m - b), bound(b) <= K m
template <class F, u32 K>
__device__ __forceinline__ void bfly_z(Fz<F>& a, Fz<F>& b) {
    const Fz<F> t = fz_add(a, b);
    b = fz_sub<F, K>(a, b);
    a = t;
}
// v < 2^6 m  ->  [0, 2 m), then [0, m) when `canonical`
template <class F>
__device__ __forceinline__ void fz_finish(Fz<F>& v, bool canonical) {
    const u32 q = v.l[8] >> 14;  // floor(v / 2^254) >= floor(v / m) >= q - 1
    const u32 qq = q ? q - 1u : 0u;
    u64 c = 0;
    i32 br = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) {
        c += (u64)qq * mod_limb<F>(i);
        const i32 d = (i32)v.l[i] - (i32)((u32)c & LIMB_MASK) + br;
        v.l[i] = (u32)d & LIMB_MASK;
        br = d >> 30;
        c >>= 30;
    }
    if (canonical) {
        Fe<F> t;
#pragma unroll
        for (int i = 0; i < NLIMBS; ++i) t.l[i] = v.l[i];
        fe_cond_sub(t);
#pragma unroll
        for (int i = 0; i < NLIMBS; ++i) v.l[i] = t.l[i];
    }
}

// inter-pass twiddles of one pass laid out as the pass reads them: d[r * Ns + k] = omega^((k r) << tw_shift), lazy form
template <class F>
__global__ void __launch_bounds__(256) ntt_direct_table_kernel(const uint4* __restrict__ z_lo, const uint4* __restrict__ z_hi, int lo_bits, uint4* __restrict__ d,
                                                               int log_ns, int s, int tw_shift) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << (log_ns + s))) return;
    const u32 k = (u32)i & ((1u << log_ns) - 1u), r = (u32)(i >> log_ns);
    Fz<F> w = twiddle_z<F>(z_lo, z_hi, (k * r) << tw_shift, lo_bits);
    u32 o[8];
    fz_store(w, o);
    d[2 * i] = make_uint4(o[0], o[1], o[2], o[3]);
    d[2 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// stage V of round 0 (rows u, u | 2^V of the thread's G = 2^LG registers)
template <class F, int LG, int V, class TW>
__device__ __forceinline__ void round0_stage_z(Fz<F> (&x)[1 << LG], const TW& tw, int s) {
    const int sh = s - 1 - V;
#pragma unroll
    for (int u = 0; u < (1 << LG); ++u) {
        if (u & (1 << V)) continue;
        const u32 ul = (u32)(u & ((1 << V) - 1));
        if (ul) {
            x[u | (1 << V)] = fz_mul(x[u | (1 << V)], tw((int)(ul << sh)));
            bfly_z<F, 2>(x[u], x[u | (1 << V)]);
        } else {
            bfly_z<F, (2u << V)>(x[u], x[u | (1 << V)]);
        }
    }
}

// in-tile twiddle table in LDS: nine limbs as they are (TWL, fits next to the data for s <= 8) or the eight memory words
template <class F, bool TWL>
struct TileTwiddles {
    uint4* a;
    uint4* b;
    u32* c;
    __device__ __forceinline__ Fz<F> operator()(int idx) const {
        if constexpr (TWL) return lds_load_limbs<F>(a, b, c, idx);
        else return lds_load_words<F>(a, b, idx);
    }
    __device__ __forceinline__ void put(int idx, const Fz<F>& v) const {
        if constexpr (TWL) lds_store_limbs<F>(a, b, c, idx, v);
        else lds_store_words<F>(a, b, idx, v);
    }
};

// stage V of a later round: rows u, u | 2^V of the thread's registers, twiddle index (L + (u mod 2^V) << stl) << sh
template <class F, int LG, int V, class TW>
__device__ __forceinline__ void round_stage_z(Fz<F> (&x)[1 << LG], const TW& tw, u32 L, int stl, int s, bool partner_zero) {
    const int sh = s - 1 - stl - V;
#pragma unroll
    for (int u = 0; u < (1 << LG); ++u) {
        if (u & (1 << V)) continue;
        if (partner_zero) {
            x[u | (1 << V)] = x[u];  // a + w * 0 = a - w * 0
        } else {
            const u32 idx = (L + ((u32)(u & ((1 << V) - 1)) << stl)) << sh;
            x[u | (1 << V)] = fz_mul(x[u | (1 << V)], tw((int)idx));  // idx 0 holds the lazy one
            bfly_z<F, 2>(x[u], x[u | (1 << V)]);
        }
    }
}

template <class F, int LG, int TLOG, bool TWL, bool FUSE>
__global__ void __launch_bounds__((1 << TLOG) >> LG) ntt_passz_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int log_n, int s, int log_ns,
                                                               const uint4* __restrict__ z_lo, const uint4* __restrict__ z_hi, int lo_bits, int last,
                                                               const uint4* __restrict__ direct, NttFusion fu) {
    constexpr int G = 1 << LG, T = 1 << TLOG, THREADS = T >> LG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = 1 << s;
    const int log_c = TLOG - s;
    const int C = 1 << log_c;
    uint4* pa = (uint4*)smem;
    uint4* pb = pa + T;
    uint4* tw_lo = pb + T;
    uint4* tw_hi = tw_lo + (R >> 1);
    u32* pc = (u32*)(tw_hi + (R >> 1));
    const TileTwiddles<F, TWL> tw{tw_lo, tw_hi, pc + T};  // word form at s = 9: exactly 80 KiB per workgroup, two per CU

    const size_t N = (size_t)1 << log_n;
    const size_t batch_off = (size_t)blockIdx.y * N * 2;
    const bool padded = FUSE && fu.in_dev;
    const size_t in_len = padded ? (size_t)1 << fu.in_log : N;  // pass 0 of a zero-padded transform reads a shorter row
    in = padded ? (const uint4*)fu.in_dev + (size_t)blockIdx.y * in_len * 2 : in + batch_off;
    out += batch_off;
    const int tid = threadIdx.x;
    const u32 c = tid & (C - 1), m = tid >> log_c;
    const u32 j = (blockIdx.x << log_c) + c;
    const u32 k = j & ((1u << log_ns) - 1u);
    const size_t row_stride = N >> s;
    // zero-padded input: rows r >= in_len / row_stride are zero.  With at most R/4 live rows a thread's rows 1..3 are
    // zero and round 0 is a broadcast; with at most R/8 the third stage is one as well (uniform over the grid)
    const u32 live_rows = (u32)(in_len / row_stride ? in_len / row_stride : 1);
    const bool bcast0 = padded && LG == 2 && s > LG && live_rows <= (u32)(R >> 2);
    const bool bcast2 = bcast0 && s >= 2 * LG && live_rows <= (u32)(R >> 3);

    Fz<F> x[G];
    const int tw_shift = log_n - log_ns - s;
    // element v of the thread (row m + v R/G) lands in the bit-reversed slot; one call per compile-time v keeps x[] in registers
    auto load_row = [&](const int v) -> Fz<F> {
        const u32 r = m + (u32)v * (u32)(R >> LG);
        const size_t idx = (size_t)j + (size_t)r * row_stride;
        Fz<F> val = fz_zero<F>();
        if (!(bcast0 && v) && idx < in_len) {
            val = load_fz<F>(in + 2 * idx);
            if (FUSE && fu.pre) val = fz_mul(val, load_fz<F>((const uint4*)fu.pre + 2 * (idx % fu.pre_period)));
            if (log_ns > 0) {
                const u32 ex = (k * r) << tw_shift;
                if (ex) val = fz_mul(val, direct ? load_fz<F>(direct + 2 * (((size_t)r << log_ns) + k)) : twiddle_z<F>(z_lo, z_hi, ex, lo_bits));
            }
        }
        return val;
    };
    if constexpr (LG == 2) {
        x[0] = load_row(0); x[2] = load_row(1); x[1] = load_row(2); x[3] = load_row(3);
    } else {
        x[0] = load_row(0); x[4] = load_row(1); x[2] = load_row(2); x[6] = load_row(3);
        x[1] = load_row(4); x[5] = load_row(5); x[3] = load_row(6); x[7] = load_row(7);
    }
    for (int i = tid; i < (R >> 1); i += THREADS) tw.put(i, twiddle_z<F>(z_lo, z_hi, (u32)i << (log_n - s), lo_bits));
    __syncthreads();

    u32 base = (s > LG) ? ((__brev(m) >> (32 - (s - LG))) << LG) : 0u;
    u32 L = 0;
    int stl = 0, vb = 0;
    // round 0: compile-time twiddles 1, w4, w8, w8^3 (index 0 is left out: operand bounds 2^(v+1))
    if (bcast0) {
#pragma unroll
        for (int u = 1; u < G; ++u) x[u] = x[0];  // a + w * 0 = a - w * 0 = a in both stages
    } else {
        round0_stage_z<F, LG, 0>(x, tw, s);
        if constexpr (LG > 1) round0_stage_z<F, LG, 1>(x, tw, s);
        if constexpr (LG > 2) round0_stage_z<F, LG, 2>(x, tw, s);
    }
    for (int st = LG; st < s; st += LG) {
#pragma unroll
        for (int u = 0; u < G; ++u) lds_store_limbs<F>(pa, pb, pc, (int)(((base + ((u32)u << stl)) << log_c) | c), x[u]);
        __syncthreads();
        stl = st + LG <= s ? st : s - LG;
        vb = st - stl;
        L = m & ((1u << stl) - 1u);
        base = L | ((m >> stl) << (stl + LG));
#pragma unroll
        for (int u = 0; u < G; ++u) x[u] = lds_load_limbs<F>(pa, pb, pc, (int)(((base + ((u32)u << stl)) << log_c) | c));
        const bool partner_zero = bcast2 && st == LG;  // third stage of a zero-padded input
        if (0 >= vb) round_stage_z<F, LG, 0>(x, tw, L, stl, s, partner_zero);
        if constexpr (LG > 1) { if (1 >= vb) round_stage_z<F, LG, 1>(x, tw, L, stl, s, false); }
        if constexpr (LG > 2) { if (2 >= vb) round_stage_z<F, LG, 2>(x, tw, L, stl, s, false); }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const u32 rr = base + ((u32)u << stl);
        const size_t dst = ((size_t)(j - k) << s) + k + ((size_t)rr << log_ns);
        if (FUSE && fu.post && last) {
            Fe<F> t;
            const Fz<F> y = fz_mul(x[u], load_fz<F>((const uint4*)fu.post + 2 * (dst % fu.post_period)));  // < m (1 + 2^-8)
#pragma unroll
            for (int i = 0; i < NLIMBS; ++i) t.l[i] = y.l[i];
            fe_cond_sub(t);
#pragma unroll
            for (int i = 0; i < NLIMBS; ++i) x[u].l[i] = t.l[i];
        } else {
            fz_finish(x[u], last != 0);
        }
        u32 w[8];
        fz_store(x[u], w);
        out[2 * dst] = make_uint4(w[0], w[1], w[2], w[3]);
        out[2 * dst + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// ---- signed lazy-domain pass (round 2; the default) ---------------------------------------------------------------------------
// The same schedule in the signed 29-bit domain of field.h (Fy, R'' = 2^261): a butterfly is one fy_mul whose multiplicand may be
// LAZY (limb-wise sums / differences, no carries), ONE carry chain for the un-multiplied operand, and two limb-wise operations
//     t = b w;   a <- norm(a);   (a, b) <- (a + t, a - t)          45 instructions beside the product instead of 63
// -- the results stay lazy through the LDS exchange (nine int32 limbs as they are) until they are either multiplied (as they are) or
// take the `a` role (normalised there).  Between passes the values travel as raw nine-limb residues in planar scratch buffers
// (16 + 16 + 4 bytes per element): the next pass multiplies every element by its inter-pass twiddle first, and a product accepts the
// lazy limbs directly, so there is no reduction, no packing and no unpacking at a pass boundary (78 instructions per element before).
// Bounds (units of m): inputs |v| < 1.13 (canonical words, a product); round 0: < 2 after the first trivial stage, < 4 after the
// second; every later stage adds a product in (-0.24, 1.24): |v| < 4 + 1.24 (s - 2) <= 12.7 for s <= 9, inside the 16 of the domain;
// lazy limbs are at most N + L1 = 1.5 * 2^30 in magnitude, and 9 * 1.5 * 2^30 * 2^29 + the reduction terms stays below 2^63.
template <class F>
__device__ __forceinline__ Fy<F> load_fy(const uint4* __restrict__ p) {
    uint4 a = p[0], b = p[1];
    return fy_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
template <class F>
__device__ __forceinline__ Fy<F> lds_load_limbs_y(const uint4* pa, const uint4* pb, const u32* pc, int idx) {
    const uint4 a = pa[idx], b = pb[idx];
    Fy<F> r;
    r.l[0] = (i32)a.x; r.l[1] = (i32)a.y; r.l[2] = (i32)a.z; r.l[3] = (i32)a.w;
    r.l[4] = (i32)b.x; r.l[5] = (i32)b.y; r.l[6] = (i32)b.z; r.l[7] = (i32)b.w;
    r.l[8] = (i32)pc[idx];
    return r;
}
template <class F>
__device__ __forceinline__ void lds_store_limbs_y(uint4* pa, uint4* pb, u32* pc, int idx, const Fy<F>& v) {
    pa[idx] = make_uint4((u32)v.l[0], (u32)v.l[1], (u32)v.l[2], (u32)v.l[3]);
    pb[idx] = make_uint4((u32)v.l[4], (u32)v.l[5], (u32)v.l[6], (u32)v.l[7]);
    pc[idx] = (u32)v.l[8];
}
template <class F>
__device__ __forceinline__ Fy<F> twiddle_y(const uint4* __restrict__ z_lo, const uint4* __restrict__ z_hi, u32 e, int lo_bits) {
    const u32 el = e & ((1u << lo_bits) - 1u), eh = e >> lo_bits;
    Fy<F> w = load_fy<F>(z_lo + 2 * (size_t)el);
    if (eh) w = fy_mul_nonneg(w, load_fy<F>(z_hi + 2 * (size_t)eh));  // stored as words: non-negative
    return w;
}
// inter-pass twiddles of one pass laid out as the pass reads them: d[r * Ns + k] = omega^((k r) << tw_shift), x 2^261 form, as raw
// BALANCED limbs in three planes ([M x 16 B][M x 16 B][M x 4 B], M = entries): the pass multiplies by them without unpacking, and
// the balanced form lets the multiplicand be the previous pass's unnormalised output (see round_stage_y)
template <class F>
__global__ void __launch_bounds__(256) ntt_direct_table_y_kernel(const uint4* __restrict__ z_lo, const uint4* __restrict__ z_hi, int lo_bits, uint4* __restrict__ d,
                                                                 int log_ns, int s, int tw_shift, uint4 sc_lo, uint4 sc_hi, int has_scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t M = (size_t)1 << (log_ns + s);
    if (i >= M) return;
    const u32 k = (u32)i & ((1u << log_ns) - 1u), r = (u32)(i >> log_ns);
    Fy<F> w = twiddle_y<F>(z_lo, z_hi, (k * r) << tw_shift, lo_bits);
    if (has_scale) {  // the constant's Montgomery words (x 2^256) read five bits lower are 32 (c 2^256) = c 2^261 + j m: the factor in this domain
        const u32 M29 = (u32)YMASK;
        Fy<F> c;
        c.l[0] = (i32)((sc_lo.x << 5) & M29);
        c.l[1] = (i32)(((sc_lo.x >> 24) | (sc_lo.y << 8)) & M29);
        c.l[2] = (i32)(((sc_lo.y >> 21) | (sc_lo.z << 11)) & M29);
        c.l[3] = (i32)(((sc_lo.z >> 18) | (sc_lo.w << 14)) & M29);
        c.l[4] = (i32)(((sc_lo.w >> 15) | (sc_hi.x << 17)) & M29);
        c.l[5] = (i32)(((sc_hi.x >> 12) | (sc_hi.y << 20)) & M29);
        c.l[6] = (i32)(((sc_hi.y >> 9) | (sc_hi.z << 23)) & M29);
        c.l[7] = (i32)(((sc_hi.z >> 6) | (sc_hi.w << 26)) & M29);
        c.l[8] = (i32)(sc_hi.w >> 3);
        w = fy_mul(c, w);
    }
    w = fy_balance(w);
    d[i] = make_uint4((u32)w.l[0], (u32)w.l[1], (u32)w.l[2], (u32)w.l[3]);
    d[M + i] = make_uint4((u32)w.l[4], (u32)w.l[5], (u32)w.l[6], (u32)w.l[7]);
    ((u32*)(d + 2 * M))[i] = (u32)w.l[8];
}
template <class F>
__device__ __forceinline__ Fy<F> load_direct_y(const uint4* __restrict__ d, size_t M, size_t i) {
    const uint4 a = d[i], b = d[M + i];
    Fy<F> r;
    r.l[0] = (i32)a.x; r.l[1] = (i32)a.y; r.l[2] = (i32)a.z; r.l[3] = (i32)a.w;
    r.l[4] = (i32)b.x; r.l[5] = (i32)b.y; r.l[6] = (i32)b.z; r.l[7] = (i32)b.w;
    r.l[8] = (i32)((const u32*)(d + 2 * M))[i];
    return r;
}
// in-tile twiddles omega_R^i (i < R / 2) of an s-stage pass as raw balanced limbs, three planes over R / 2 entries: what a 9-stage
// pass copies its LDS table from and reads its last stage's factors from (TWM 2 below)
template <class F>
__global__ void __launch_bounds__(256) ntt_tile_table_y_kernel(const uint4* __restrict__ z_lo, const uint4* __restrict__ z_hi, int lo_bits, uint4* __restrict__ d,
                                                               int log_n, int s) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 M = 1u << (s - 1);
    if (i >= M) return;
    const Fy<F> w = fy_balance(twiddle_y<F>(z_lo, z_hi, i << (log_n - s), lo_bits));
    d[i] = make_uint4((u32)w.l[0], (u32)w.l[1], (u32)w.l[2], (u32)w.l[3]);
    d[M + i] = make_uint4((u32)w.l[4], (u32)w.l[5], (u32)w.l[6], (u32)w.l[7]);
    ((u32*)(d + 2 * M))[i] = (u32)w.l[8];
}
// in-tile twiddle table of the signed passes.  TWM 0: the eight packed words of R / 2 entries (unsigned, unpacked on every use);
// TWM 1: R / 2 entries as nine balanced limbs (fits beside the data for s <= 8); TWM 2 (s = 9): the R / 4 EVEN-index entries as
// limbs -- every stage but the last reads even indices only -- and the last stage's factors straight from the 9 KiB global table g
// (two per thread, L1-resident): 76.5 KiB per workgroup, so the 9-stage pass gets the balanced schedule of round_stage_y as well
template <class F, int TWM>
struct TileTwiddlesY {
    static constexpr bool HALF = TWM == 2;
    uint4* a;
    uint4* b;
    u32* c;
    const uint4* g;
    int half_r;
    __device__ __forceinline__ Fy<F> operator()(int idx) const {
        if constexpr (TWM == 2) return lds_load_limbs_y<F>(a, b, c, idx >> 1);
        else if constexpr (TWM == 1) return lds_load_limbs_y<F>(a, b, c, idx);
        else { const uint4 x = a[idx], y = b[idx]; return fy_load<F>(x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w); }
    }
    __device__ __forceinline__ Fy<F> tail(int idx) const { return load_direct_y<F>(g, (size_t)half_r, (size_t)idx); }
    __device__ __forceinline__ void put(int idx, const Fy<F>& v) const {
        if constexpr (TWM != 0) lds_store_limbs_y<F>(a, b, c, idx, fy_balance(v));  // balanced: see round_stage_y
        else { u32 w[8]; fy_store(v, w); a[idx] = make_uint4(w[0], w[1], w[2], w[3]); b[idx] = make_uint4(w[4], w[5], w[6], w[7]); }
    }
};
// Limb growth and where the butterflies normalise.  A butterfly is x[u] = a + t, x[u'] = a - t with t = x[u'] w a fresh product
// (normalised limbs) and a = x[u] carried over: limb-wise sums, so |limb of a| grows by 2^29 per stage ("level").  Values only
// have to fit 32 bits (level <= 3) -- PROVIDED the twiddle is in the balanced form (BAL: |limb| <= 2^28, the LDS limb table and
// the direct tables): a column of multiplicand x twiddle is then < 9 * 3 * 2^29 * 2^28 + reduction terms < 2^62.  With the packed
// (unsigned, < 2^29) twiddles of 9-stage passes the multiplicand must stay below level 2, i.e. every a is normalised first.
// BAL schedule for the four rows of a thread (two stages per round): loaded values have level 1; round 0 runs without any
// normalisation (level 3 at its end); every later round normalises the carried operands of its FIRST stage only (rows 0 and 2:
// level 3 -> 1, results level 2), its second stage carries level 2 -> 3.  Two carry chains per round and thread instead of four.
template <class F, int LG, int V, bool BAL, class TW>
__device__ __forceinline__ void round0_stage_y(Fy<F> (&x)[1 << LG], const TW& tw, int s) {
    const int sh = s - 1 - V;
#pragma unroll
    for (int u = 0; u < (1 << LG); ++u) {
        if (u & (1 << V)) continue;
        const u32 ul = (u32)(u & ((1 << V) - 1));
        Fy<F> t = x[u | (1 << V)];
        if (ul) t = fy_mul(t, tw((int)(ul << sh)));      // lazy multiplicand
        const Fy<F> a = (V && !(BAL && LG == 2)) ? fy_norm(x[u]) : x[u];  // stage 0 sees the loaded values; later ones the previous stage's lazy sums
        x[u] = fy_add_lazy(a, t);
        x[u | (1 << V)] = fy_sub_lazy(a, t);
    }
}
// stage V of a later round; the values come from the LDS exchange (lazy) or from the stage before (lazy).  first: the first stage
// this round executes
template <class F, int LG, int V, bool BAL, class TW>
__device__ __forceinline__ void round_stage_y(Fy<F> (&x)[1 << LG], const TW& tw, u32 L, int stl, int s, bool partner_zero, bool first) {
    const int sh = s - 1 - stl - V;
#pragma unroll
    for (int u = 0; u < (1 << LG); ++u) {
        if (u & (1 << V)) continue;
        if (partner_zero) {
            x[u | (1 << V)] = x[u];  // a + w * 0 = a - w * 0
        } else {
            const u32 idx = (L + ((u32)(u & ((1 << V) - 1)) << stl)) << sh;
            Fy<F> w;
            if (TW::HALF && sh == 0) w = tw.tail((int)idx);  // the pass's last stage: odd indices too
            else w = tw((int)idx);                          // idx 0 holds the lazy one
            const Fy<F> t = fy_mul(x[u | (1 << V)], w);
            const Fy<F> a = (!(BAL && LG == 2) || first) ? fy_norm(x[u]) : x[u];
            x[u] = fy_add_lazy(a, t);
            x[u | (1 << V)] = fy_sub_lazy(a, t);
        }
    }
}
// normalised |v| < 16 m  ->  the canonical residue in [0, m) as eight words.  q = floor(v / 2^254); v - (q - [q >= 0]) m lies in [0, 2 m)
// for every such v (m = 2^254 + t, t < 2^126), so one conditional subtraction finishes
template <class F>
__device__ __forceinline__ void fy_canonical_words(const Fy<F>& v, u32* w) {
    const i32 q = v.l[8] >> 22;
    const i32 mult = q - (q >= 0 ? 1 : 0);
    Fy<F> r;
    i64 c = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS - 1; ++i) {
        c += (i64)v.l[i] - (i64)mult * ymod_limb<F>(i);
        r.l[i] = (i32)((u32)c & (u32)YMASK);
        c >>= YBITS;
    }
    r.l[8] = (i32)(c + v.l[8] - (i64)mult * ymod_limb<F>(8));
    // conditional subtraction of m
    i32 t[NLIMBS], b = 0;
#pragma unroll
    for (int i = 0; i < NLIMBS - 1; ++i) {
        const i32 d = r.l[i] - ymod_limb<F>(i) + b;
        t[i] = d & YMASK;
        b = d >> YBITS;
    }
    t[8] = r.l[8] - ymod_limb<F>(8) + b;
    const bool ge = t[8] >= 0;
#pragma unroll
    for (int i = 0; i < NLIMBS; ++i) r.l[i] = ge ? t[i] : r.l[i];
    fy_store(r, w);
}

// in / out: eight-word elements (the caller's buffer: pass 0 input, last pass output) or raw nine-limb planes of the scratch
// (raw_in / raw_out: [N x 16 B][N x 16 B][N x 4 B] per transform)
template <class F, int LG, int TLOG, int TWM, bool FUSE>
__global__ void __launch_bounds__((1 << TLOG) >> LG) ntt_passy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int log_n, int s, int log_ns,
                                                               const uint4* __restrict__ z_lo, const uint4* __restrict__ z_hi, int lo_bits, int last, int raw_in, int raw_out,
                                                               const uint4* __restrict__ direct, NttFusion fu, int batch_major, const uint4* __restrict__ tile_tab) {
    constexpr bool TWL = TWM != 0;  // balanced limb twiddles: the schedule with half the normalisations
    constexpr int G = 1 << LG, T = 1 << TLOG;
    // grid order: tile-major (x = tile, y = transform) or batch-major (x = transform, y = tile: consecutive workgroups run the SAME
    // tile of consecutive transforms, so the rows of the shared inter-pass twiddle table they read stay in L2)
    const u32 bx = batch_major ? blockIdx.y : blockIdx.x, by = batch_major ? blockIdx.x : blockIdx.y;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = 1 << s;
    const int log_c = TLOG - s;
    const int C = 1 << log_c;
    uint4* pa = (uint4*)smem;
    uint4* pb = pa + T;
    const int tw_n = TWM == 2 ? (R >> 2) : (R >> 1);
    uint4* tw_lo = pb + T;
    uint4* tw_hi = tw_lo + tw_n;
    u32* pc = (u32*)(tw_hi + tw_n);
    const TileTwiddlesY<F, TWM> tw{tw_lo, tw_hi, pc + T, tile_tab, R >> 1};

    const size_t N = (size_t)1 << log_n;
    const bool padded = FUSE && fu.in_dev;
    const size_t in_len = padded ? (size_t)1 << fu.in_log : N;
    // element strides of a transform: 2 N uint4 in word form, 36 N bytes = 2 N uint4 + N u32 in raw form
    const u32 blk = (FUSE && fu.blocks) ? by % fu.blocks : 0u;           // coset block of this transform
    const size_t in_row = (FUSE && fu.pre_blocks) ? by / fu.blocks : by;  // the blocks of one polynomial share its coefficients
    const uint4* in_a = padded ? (const uint4*)fu.in_dev + in_row * in_len * 2 : raw_in ? (const uint4*)((const char*)in + (size_t)by * N * 36) : in + (size_t)by * N * 2;
    uint4* out_a = raw_out ? (uint4*)((char*)out + (size_t)by * N * 36) : out + (size_t)by * N * 2;
    const int tid = threadIdx.x;
    const u32 c = tid & (C - 1), m = tid >> log_c;
    const u32 j = (bx << log_c) + c;
    const u32 k = j & ((1u << log_ns) - 1u);
    const size_t row_stride = N >> s;
    const u32 live_rows = (u32)(in_len / row_stride ? in_len / row_stride : 1);
    const bool bcast0 = padded && LG == 2 && s > LG && live_rows <= (u32)(R >> 2);
    const bool bcast2 = bcast0 && s >= 2 * LG && live_rows <= (u32)(R >> 3);

    Fy<F> x[G];
    const int tw_shift = log_n - log_ns - s;
    auto load_row = [&](const int v) -> Fy<F> {
        const u32 r = m + (u32)v * (u32)(R >> LG);
        const size_t idx = (size_t)j + (size_t)r * row_stride;
        Fy<F> val = fy_zero<F>();
        if (!(bcast0 && v) && idx < in_len) {
            if (raw_in) {
                const uint4 a = in_a[idx], b = in_a[N + idx];
                val.l[0] = (i32)a.x; val.l[1] = (i32)a.y; val.l[2] = (i32)a.z; val.l[3] = (i32)a.w;
                val.l[4] = (i32)b.x; val.l[5] = (i32)b.y; val.l[6] = (i32)b.z; val.l[7] = (i32)b.w;
                val.l[8] = (i32)((const u32*)(in_a + 2 * N))[idx];
            } else {
                val = load_fy<F>(in_a + 2 * idx);
            }
            if (FUSE && fu.pre) val = fy_mul(val, load_fy<F>((const uint4*)fu.pre + 2 * (idx % fu.pre_period)));
            if (FUSE && fu.pre_blocks && !raw_in) val = fy_mul(val, load_direct_y<F>((const uint4*)fu.pre_blocks, (size_t)fu.blocks << log_n, ((size_t)blk << log_n) + idx));
            if (log_ns > 0)  // every element of a later pass: the product is also what brings a raw residue back to |v| < 1.13 m
                val = fy_mul(val, direct ? load_direct_y<F>(direct, (size_t)1 << (log_ns + s), ((size_t)r << log_ns) + k) : fy_balance(twiddle_y<F>(z_lo, z_hi, (k * r) << tw_shift, lo_bits)));
        }
        return val;
    };
    if constexpr (LG == 2) {
        x[0] = load_row(0); x[2] = load_row(1); x[1] = load_row(2); x[3] = load_row(3);
    } else {
        x[0] = load_row(0); x[4] = load_row(1); x[2] = load_row(2); x[6] = load_row(3);
        x[1] = load_row(4); x[5] = load_row(5); x[3] = load_row(6); x[7] = load_row(7);
    }
    if constexpr (TWM == 2) {
        for (int i = tid; i < tw_n; i += (T >> LG)) lds_store_limbs_y<F>(tw_lo, tw_hi, pc + T, i, load_direct_y<F>(tile_tab, (size_t)(R >> 1), (size_t)(2 * i)));
    } else {
        for (int i = tid; i < tw_n; i += (T >> LG)) tw.put(i, twiddle_y<F>(z_lo, z_hi, (u32)i << (log_n - s), lo_bits));
    }
    __syncthreads();

    u32 base = (s > LG) ? ((__brev(m) >> (32 - (s - LG))) << LG) : 0u;
    u32 L = 0;
    int stl = 0, vb = 0;
    if (bcast0) {
#pragma unroll
        for (int u = 1; u < G; ++u) x[u] = x[0];
    } else {
        round0_stage_y<F, LG, 0, TWL>(x, tw, s);
        if constexpr (LG > 1) round0_stage_y<F, LG, 1, TWL>(x, tw, s);
        if constexpr (LG > 2) round0_stage_y<F, LG, 2, TWL>(x, tw, s);
    }
    for (int st = LG; st < s; st += LG) {
#pragma unroll
        for (int u = 0; u < G; ++u) lds_store_limbs_y<F>(pa, pb, pc, (int)(((base + ((u32)u << stl)) << log_c) | c), x[u]);
        __syncthreads();
        stl = st + LG <= s ? st : s - LG;
        vb = st - stl;
        L = m & ((1u << stl) - 1u);
        base = L | ((m >> stl) << (stl + LG));
#pragma unroll
        for (int u = 0; u < G; ++u) x[u] = lds_load_limbs_y<F>(pa, pb, pc, (int)(((base + ((u32)u << stl)) << log_c) | c));
        const bool partner_zero = bcast2 && st == LG;
        if (0 >= vb) round_stage_y<F, LG, 0, TWL>(x, tw, L, stl, s, partner_zero, true);
        if constexpr (LG > 1) { if (1 >= vb) round_stage_y<F, LG, 1, TWL>(x, tw, L, stl, s, false, vb == 1); }
        if constexpr (LG > 2) { if (2 >= vb) round_stage_y<F, LG, 2, TWL>(x, tw, L, stl, s, false, vb == 2); }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const u32 rr = base + ((u32)u << stl);
        const size_t dst = ((size_t)(j - k) << s) + k + ((size_t)rr << log_ns);
        if (raw_out) {  // lazy limbs as they are: the next pass's product takes them
            out_a[dst] = make_uint4((u32)x[u].l[0], (u32)x[u].l[1], (u32)x[u].l[2], (u32)x[u].l[3]);
            out_a[N + dst] = make_uint4((u32)x[u].l[4], (u32)x[u].l[5], (u32)x[u].l[6], (u32)x[u].l[7]);
            ((u32*)(out_a + 2 * N))[dst] = (u32)x[u].l[8];
        } else {
            Fy<F> y;
            if (FUSE && fu.post && last) y = fy_mul(fy_norm(x[u]), load_fy<F>((const uint4*)fu.post + 2 * (dst % fu.post_period)));  // packed factor: the multiplicand must be normalised
            else if (FUSE && fu.post_blocks && last) y = fy_mul(x[u], load_direct_y<F>((const uint4*)fu.post_blocks, (size_t)fu.blocks << log_n, ((size_t)blk << log_n) + dst));  // balanced factor: lazy limbs as they are
            else y = fy_norm(x[u]);
            u32 w[8];
            fy_canonical_words(y, w);
            out_a[2 * dst] = make_uint4(w[0], w[1], w[2], w[3]);
            out_a[2 * dst + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
}

// ---- persistent, software-pipelined form of the signed pass (round 5) ---------------------------------------------------------------
// ntt_passy_kernel pays ~30 % of its time for memory it has nothing to overlap with: every workgroup starts by waiting for its own
// loads and ends behind its own stores (NOTEBOOK "(f)": 2^22 0.470 ms, 0.319 with no memory operations).  Here a workgroup stays
// resident and walks over tiles w = blockIdx.x, + gridDim.x, ...; while the LAST round of tile t computes, tile t + 1 is already on its way:
//   * the elements by LDS-DMA (global_load_lds_dwordx4: no VGPRs) straight into the exchange buffer, which is free from the moment
//     every wave has made the last exchange read of tile t.  The DMA puts global row r of the tile at LDS row bitrev_s(r): exactly the
//     slots thread (m, c) owns in round 0 (rows (bitrev(m) << 2) + u), so round 0 reads its four elements from LDS instead of HBM and
//     writes its results back to the same slots -- no extra exchange;
//   * the per-element factor (inter-pass twiddle of passes >= 1, or the coset-block table of coeff_to_extended_blocks on pass 0) into
//     36 registers by ordinary loads, issued at the same point (the last round holds four elements + these + a product's temporaries,
//     the same pressure the non-persistent kernel has in its round 0);
//   * the stores of tile t are issued after the barrier that publishes tile t + 1's DMA and drain under round 0 of tile t + 1.
// The in-tile twiddle table is built once per workgroup instead of once per tile.  Not taken here (ntt_passy_kernel keeps them):
// zero-padded inputs, the packed periodic factors (pre / post), passes without a direct table, s < 4.
struct NttPipeArgs {
    const uint4* in; uint4* out;
    int log_n, s, log_ns, last, words_in, raw_out;
    const uint4* fac;        // factor table in the three-plane balanced form (null: none)
    unsigned long long fac_M;  // its entries per plane
    u32 fac_blocks;          // 0: inter-pass table, entry (r << log_ns) + k; > 0: coset-block table, entry ((t % blocks) << log_n) + index
    u32 in_div;              // word-form input: transform t reads row t / in_div (the blocks of one polynomial share its coefficients)
    const uint4* post_blocks; u32 blocks;
    u32 nb; int batch_major;
    const uint4* tile_tab; const uint4* z_lo; const uint4* z_hi; int lo_bits;
};
// 16 bytes per lane from global address sbase + voff (sbase wave-uniform in SGPRs, voff the lane's 32-bit byte offset) to LDS byte address
// lds_dst + lane * 16 (lds_dst wave-uniform, through M0)
__device__ __forceinline__ void glds16(const void* sbase, u32 voff, u32 lds_dst) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// MODE bit 0: word-form input (pass 0); bit 1: a factor table; bit 2: coset-block factors on the final store (post_blocks); bit 3: raw output
template <class F, int TWM, int MODE>
__global__ void __launch_bounds__(TILE >> 2) __attribute__((amdgpu_waves_per_eu(4, 4))) ntt_passp_kernel(const NttPipeArgs p) {
    constexpr int LG = 2, G = 4, T = TILE, TLOG = TILE_LOG;
    constexpr bool WORDS_IN = (MODE & 1) != 0, HAS_FAC = (MODE & 2) != 0, POSTB = (MODE & 4) != 0, RAW_OUT = (MODE & 8) != 0;
#ifndef TRH_NTT_PF
#define TRH_NTT_PF 1
#endif
#ifndef TRH_NTT_PFMODE
#define TRH_NTT_PFMODE 0  // 1: the other factor rows are requested before the tile's stores; 0: at the start of the next tile
#endif
    constexpr int PF = TRH_NTT_PF;  // factor rows fetched a round ahead (the others at the start of the tile)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = p.s, log_n = p.log_n, log_ns = p.log_ns;
    const int R = 1 << s;
    const int log_c = TLOG - s;
    const int C = 1 << log_c;
    uint4* pa = (uint4*)smem;
    uint4* pb = pa + T;
    const int tw_n = TWM == 2 ? (R >> 2) : (R >> 1);
    uint4* tw_lo = pb + T;
    uint4* tw_hi = tw_lo + tw_n;
    u32* pc = (u32*)(tw_hi + tw_n);
    const TileTwiddlesY<F, TWM> tw{tw_lo, tw_hi, pc + T, p.tile_tab, R >> 1};
    const size_t N = (size_t)1 << log_n;
    const int tid = threadIdx.x;
    const u32 c0 = tid & (C - 1), m0 = tid >> log_c;
    const size_t row_stride = N >> s;
    const u32 tiles_log = (u32)(log_n - TLOG);
    const u32 total = p.nb << tiles_log;
    const u32 lds_a = (u32)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const u32 lds_b = lds_a + T * 16, lds_c = lds_a + 2 * T * 16 + 2 * (u32)tw_n * 16;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane(tid >> 6), lane = (u32)tid & 63u;

    auto coords = [&](u32 w, u32& bx, u32& by) {
        if (p.batch_major) { bx = w / p.nb; by = w - bx * p.nb; }
        else { by = w >> tiles_log; bx = w & ((1u << tiles_log) - 1u); }
    };
    // a tile's DMA: the lane offsets (row bitrev_s(slot row), column) are the same for every tile; the tile and the transform move the
    // wave-uniform base
    auto issue_dma = [&](u32 bx, u32 by) {
        const char* src = WORDS_IN ? (const char*)(p.in + (size_t)(by / p.in_div) * N * 2) : (const char*)p.in + (size_t)by * N * 36;
        src += ((size_t)bx << log_c) * (WORDS_IN ? 32 : 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 e0 = (wv * 4 + (u32)i) * 64u;  // wave-uniform first slot of this instruction
            const u32 e = e0 + lane;
            const u32 rho = e >> log_c, cc = e & (u32)(C - 1);
            const u32 r = __brev(rho) >> (32 - s);
            const u32 g = cc + r * (u32)row_stride;
            const u32 da = (u32)__builtin_amdgcn_readfirstlane((int)(lds_a + e0 * 16u)), db = (u32)__builtin_amdgcn_readfirstlane((int)(lds_b + e0 * 16u));
            if constexpr (WORDS_IN) { glds16(src, g * 32u, da); glds16(src + 16, g * 32u, db); }
            else { glds16(src, g * 16u, da); glds16(src + N * 16, g * 16u, db); }
        }
        if constexpr (!WORDS_IN) {  // the 4-byte plane: four consecutive elements of a row (C >= 4) per lane
            const u32 e0 = wv * 256u;
            const u32 e = e0 + lane * 4u;
            const u32 rho = e >> log_c, cc = e & (u32)(C - 1);
            const u32 r = __brev(rho) >> (32 - s);
            const u32 g = cc + r * (u32)row_stride;
            glds16(src + N * 32 - ((size_t)bx << log_c) * 12, g * 4u, (u32)__builtin_amdgcn_readfirstlane((int)(lds_c + e0 * 4u)));
        }
    };
    // factor of the element x[u] holds (row m + bitrev2(u) R / 4 of the tile).  Rows u = 0, 1 are fetched a round ahead (18 registers beside the
    // last round's four elements); rows 2, 3 at the start of the tile, under the products of rows 0, 1 (36 more would spill)
    auto load_fac = [&](u32 bx, u32 by, int u, u32 m, u32 c) -> Fy<F> {
        const u32 j = (bx << log_c) + c;
        const u32 r = m + (u32)(((u & 1) << 1) | (u >> 1)) * (u32)(R >> LG);
        const size_t idx = p.fac_blocks ? ((size_t)(by % p.fac_blocks) << log_n) + (size_t)j + (size_t)r * row_stride : ((size_t)r << log_ns) + (j & ((1u << log_ns) - 1u));
        return load_direct_y<F>(p.fac, (size_t)p.fac_M, idx);
    };

    u32 w = blockIdx.x, bx = 0, by = 0;
    Fy<F> f0 = fy_zero<F>(), f1 = fy_zero<F>(), f2 = fy_zero<F>(), f3 = fy_zero<F>();
    if (w < total) {
        coords(w, bx, by);
        issue_dma(bx, by);
        if constexpr (HAS_FAC) { f0 = load_fac(bx, by, 0, m0, c0); f1 = load_fac(bx, by, 1, m0, c0); f2 = load_fac(bx, by, 2, m0, c0); f3 = load_fac(bx, by, 3, m0, c0); }
    }
    if constexpr (TWM == 2) {
        for (int i = tid; i < tw_n; i += (T >> LG)) lds_store_limbs_y<F>(tw_lo, tw_hi, pc + T, i, load_direct_y<F>(p.tile_tab, (size_t)(R >> 1), (size_t)(2 * i)));
    } else {
        for (int i = tid; i < tw_n; i += (T >> LG)) tw.put(i, twiddle_y<F>(p.z_lo, p.z_hi, (u32)i << (log_n - s), p.lo_bits));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    while (w < total) {
        // (m, c) pass through an empty asm statement once per tile: otherwise the compiler keeps the dozen 64-bit factor / output addresses that
        // depend on them alone live across the whole loop and spills them (76 - 96 bytes of scratch per thread, reloaded behind vmcnt(0))
        u32 m = m0, c = c0;
        asm volatile("" : "+v"(m), "+v"(c));
        const u32 j = (bx << log_c) + c;
        const u32 k = j & ((1u << log_ns) - 1u);
        const size_t blk = (POSTB && p.blocks) ? by % p.blocks : 0u;
        uint4* out_a = RAW_OUT ? (uint4*)((char*)p.out + (size_t)by * N * 36) : p.out + (size_t)by * N * 2;
        u32 base = (__brev(m) >> (32 - (s - LG))) << LG;
        Fy<F> x[G];
        // round 0: the thread's own four slots, filled by the DMA
        auto own = [&](int u) -> Fy<F> {
            const int idx = (int)(((base + (u32)u) << log_c) | c);
            if constexpr (WORDS_IN) { const uint4 a = pa[idx], b = pb[idx]; return fy_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w); }
            else return lds_load_limbs_y<F>(pa, pb, pc, idx);
        };
        if constexpr (HAS_FAC && TRH_NTT_PFMODE == 0) { if constexpr (PF < 2) f1 = load_fac(bx, by, 1, m, c); f2 = load_fac(bx, by, 2, m, c); f3 = load_fac(bx, by, 3, m, c); }
        x[0] = own(0); x[1] = own(1);
        if constexpr (HAS_FAC) { x[0] = fy_mul(x[0], f0); x[1] = fy_mul(x[1], f1); }
        x[2] = own(2); x[3] = own(3);
        if constexpr (HAS_FAC) { x[2] = fy_mul(x[2], f2); x[3] = fy_mul(x[3], f3); }
        round0_stage_y<F, LG, 0, true>(x, tw, s);
        round0_stage_y<F, LG, 1, true>(x, tw, s);
        const u32 wn = w + gridDim.x;
        const bool has_next = wn < total;
        u32 nbx = 0, nby = 0;
        if (has_next) coords(wn, nbx, nby);
        u32 L = 0;
        int stl = 0, vb = 0;
        for (int st = LG; st < s; st += LG) {
#pragma unroll
            for (int u = 0; u < G; ++u) lds_store_limbs_y<F>(pa, pb, pc, (int)(((base + ((u32)u << stl)) << log_c) | c), x[u]);
            __syncthreads();
            stl = st + LG <= s ? st : s - LG;
            vb = st - stl;
            L = m & ((1u << stl) - 1u);
            base = L | ((m >> stl) << (stl + LG));
#pragma unroll
            for (int u = 0; u < G; ++u) x[u] = lds_load_limbs_y<F>(pa, pb, pc, (int)(((base + ((u32)u << stl)) << log_c) | c));
            if (st + LG >= s) {  // the last exchange read: once every wave has made it the buffer takes the next tile
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (has_next) {
                    issue_dma(nbx, nby);
                    if constexpr (HAS_FAC) { f0 = load_fac(nbx, nby, 0, m, c); if constexpr (PF >= 2) f1 = load_fac(nbx, nby, 1, m, c); }
                }
            }
            if (0 >= vb) round_stage_y<F, LG, 0, true>(x, tw, L, stl, s, false, true);
            if (1 >= vb) round_stage_y<F, LG, 1, true>(x, tw, L, stl, s, false, vb == 1);
        }
        // the next tile's elements have landed (this wave's; the barrier makes it every wave's) before the stores go out: they drain
        // under the next tile's round 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // the remaining factor rows of the next tile go out BEFORE the stores: the memory counter retires in order, so a load issued behind
        // the stores could not be waited for without waiting for them
        if constexpr (HAS_FAC && TRH_NTT_PFMODE == 1) {
            if (has_next) { if constexpr (PF < 2) f1 = load_fac(nbx, nby, 1, m, c); f2 = load_fac(nbx, nby, 2, m, c); f3 = load_fac(nbx, nby, 3, m, c); }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const u32 rr = base + ((u32)u << stl);
            const size_t dst = ((size_t)(j - k) << s) + k + ((size_t)rr << log_ns);
            if constexpr (RAW_OUT) {
                out_a[dst] = make_uint4((u32)x[u].l[0], (u32)x[u].l[1], (u32)x[u].l[2], (u32)x[u].l[3]);
                out_a[N + dst] = make_uint4((u32)x[u].l[4], (u32)x[u].l[5], (u32)x[u].l[6], (u32)x[u].l[7]);
                ((u32*)(out_a + 2 * N))[dst] = (u32)x[u].l[8];
            } else {
                Fy<F> y;
                if (POSTB && p.post_blocks && p.last) y = fy_mul(x[u], load_direct_y<F>(p.post_blocks, (size_t)p.blocks << log_n, (blk << log_n) + dst));
                else y = fy_norm(x[u]);
                u32 wd[8];
                fy_canonical_words(y, wd);
                out_a[2 * dst] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
                out_a[2 * dst + 1] = make_uint4(wd[4], wd[5], wd[6], wd[7]);
            }
        }
        w = wn; bx = nbx; by = nby;
    }
}

template <class F>
__global__ void __launch_bounds__(256) field_scale_periodic_kernel(uint4* __restrict__ a, size_t rows, size_t row_len, size_t active_len,
                                                                   const uint4* __restrict__ factors, u32 period) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * active_len) return;
    const size_t r = i / active_len, c = i - r * active_len;
    uint4* p = a + 2 * (r * row_len + c);
    Fe<F> f = load_fe<F>(factors + 2 * (c % period));
    store_fe<F>(p, fe_mul(load_fe<F>(p), f));
}

// table[b][i] = (g w^b)^i * scale as raw balanced limbs of the x 2^261 form, three planes over M = blocks << log_n entries
template <class F>
__global__ void __launch_bounds__(256) ntt_block_table_kernel(uint4* __restrict__ d, u32 blocks, int log_n, const uint4* __restrict__ consts /* g, w, scale */) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t M = (size_t)blocks << log_n;
    if (i >= M) return;
    const u32 b = (u32)(i >> log_n), e = (u32)i & ((1u << log_n) - 1u);
    Fe<F> base = load_fe<F>(consts);
    const Fe<F> w = load_fe<F>(consts + 2);
    for (u32 t = 0; t < b; ++t) base = fe_mul(base, w);  // b < 64
    Fe<F> r = load_fe<F>(consts + 4);
    for (int bit = 0; bit < log_n; ++bit) {
        if ((e >> bit) & 1u) r = fe_mul(r, base);
        base = fe_sqr(base);
    }
    const Fy<F> v = fy_balance(fy_from_fe(r));
    d[i] = make_uint4((u32)v.l[0], (u32)v.l[1], (u32)v.l[2], (u32)v.l[3]);
    d[M + i] = make_uint4((u32)v.l[4], (u32)v.l[5], (u32)v.l[6], (u32)v.l[7]);
    ((u32*)(d + 2 * M))[i] = (u32)v.l[8];
}
// out[t][i] = in[in_per_block ? t : t / blocks][i] * table[t % blocks][i], canonical words
template <class F>
__global__ void __launch_bounds__(256) ntt_block_scale_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t transforms, u32 blocks, int log_n,
                                                              const uint4* __restrict__ table, int in_per_block) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (transforms << log_n)) return;
    const size_t t = i >> log_n, e = i & (((size_t)1 << log_n) - 1);
    const size_t src = ((in_per_block ? t : t / blocks) << log_n) + e;
    const Fy<F> y = fy_mul(load_fy<F>(in + 2 * src), load_direct_y<F>(table, (size_t)blocks << log_n, ((t % blocks) << log_n) + e));
    u32 w[8];
    fy_canonical_words(y, w);
    out[2 * i] = make_uint4(w[0], w[1], w[2], w[3]);
    out[2 * i + 1] = make_uint4(w[4], w[5], w[6], w[7]);
}

TwiddleEntry* find_tables(int field, int log_n, const u64 omega[4], const u64* scale) {
    Ctx& c = ctx();
    for (TwiddleEntry* t : c.twiddles)
        if (t->field == field && t->log_n == log_n && memcmp(t->omega, omega, 32) == 0 && t->has_scale == (scale != nullptr) &&
            (!scale || memcmp(t->scale, scale, 32) == 0)) { t->stamp = ++c.stamp; return t; }
    return nullptr;
}

// pass plan: log_n split into passes of <= MAX_PASS_LOG stages on 2^tlog-element tiles
void plan_passes(int log_n, int* sizes, int* n_passes, int* tile_log) {
    // (a 4096-element tile -- two passes for 2^19..2^22, all 160 KiB of LDS, one workgroup per CU -- measured equal: removed)
    int P = 0, tlog = TILE_LOG;
    if (log_n <= TILE_LOG) { sizes[P++] = log_n; }
    else {
        P = (log_n + MAX_PASS_LOG - 1) / MAX_PASS_LOG;
        if (P < 2) P = 2;
        int rem = log_n;
        for (int p = 0; p < P; ++p) { sizes[p] = (rem + (P - p) - 1) / (P - p); rem -= sizes[p]; }
        if (const char* e = getenv("TRH_NTT_PLAN")) {  // tuning knob: explicit pass sizes "8,8,6"
            int v[8], cnt = 0, sum = 0;
            for (const char* q = e; *q && cnt < 8;) { v[cnt] = atoi(q); sum += v[cnt++]; while (*q && *q != ',') ++q; if (*q == ',') ++q; }
            if (sum == log_n) { P = cnt; for (int p = 0; p < P; ++p) sizes[p] = v[p]; }
        }
    }
    *n_passes = P; *tile_log = tlog;
}
bool lazy_enabled() {
    static const int lazy = getenv("TRH_NTT_LAZY") ? atoi(getenv("TRH_NTT_LAZY")) : 1;
    return lazy != 0;
}
// the signed 29-bit lazy passes (ntt_passy_kernel) instead of the unsigned 30-bit ones (ntt_passz_kernel); TRH_NTT_SIGNED=0 for A/B
bool signed_enabled() {
    static const int sg = getenv("TRH_NTT_SIGNED") ? atoi(getenv("TRH_NTT_SIGNED")) : 1;
    return sg != 0 && lazy_enabled();
}

template <class F>
int build_tables(int log_n, const u64 omega[4], const u64* scale, hipStream_t s, TwiddleEntry** out) {
    Ctx& c = ctx();
    if (c.twiddles.size() >= 16) {  // evict the least recently used entry
        size_t victim = 0;
        for (size_t i = 1; i < c.twiddles.size(); ++i)
            if (c.twiddles[i]->stamp < c.twiddles[victim]->stamp) victim = i;
        TRH_HIP_TRY(hipDeviceSynchronize());
        c.twiddles[victim]->release_all();
        delete c.twiddles[victim];
        c.twiddles.erase(c.twiddles.begin() + victim);
    }
    TwiddleEntry* t = new TwiddleEntry();
    t->field = F::ID; t->log_n = log_n; memcpy(t->omega, omega, 32);
    if (scale) { t->has_scale = true; memcpy(t->scale, scale, 32); }
    t->lo_bits = (log_n + 1) / 2; t->hi_bits = log_n - t->lo_bits;
    if (t->lo_bits < 1) t->lo_bits = 1;
    t->stamp = ++c.stamp;
    int rc = t->lo.ensure(((size_t)32 << t->lo_bits) + 32 * 64);
    if (rc == TRH_OK) rc = t->hi.ensure((size_t)32 << t->hi_bits);
    if (rc == TRH_OK) rc = t->zlo.ensure((size_t)32 << t->lo_bits);
    if (rc == TRH_OK) rc = t->zhi.ensure((size_t)32 << t->hi_bits);
    if (rc != TRH_OK) { delete t; return rc; }
    // omega^(2^b) on the host (shared field code), staged behind the lo table
    FeMem pw[32];
    memcpy(&pw[0], omega, 32);
    for (int b = 1; b < 32; ++b) fe_store(fe_sqr(fe_load<F>(pw[b - 1])), pw[b]);
    uint4* d_pw = t->lo.as<uint4>() + ((size_t)2 << t->lo_bits);
    TRH_HIP_TRY(hipMemcpyAsync(d_pw, pw, sizeof(pw), hipMemcpyHostToDevice, s));
    const u32 cnt = 1u << (t->lo_bits > t->hi_bits ? t->lo_bits : t->hi_bits);
    hipLaunchKernelGGL((ntt_tables_kernel<F>), dim3((cnt + 255) / 256), dim3(256), 0, s, d_pw, t->lo.as<uint4>(), t->hi.as<uint4>(), t->lo_bits, t->hi_bits);
    if (signed_enabled()) {
        hipLaunchKernelGGL((ntt_tables_lazy_kernel<F, true>), dim3(((1u << t->lo_bits) + 255) / 256), dim3(256), 0, s, t->lo.as<uint4>(), t->zlo.as<uint4>(), 1u << t->lo_bits);
        hipLaunchKernelGGL((ntt_tables_lazy_kernel<F, true>), dim3(((1u << t->hi_bits) + 255) / 256), dim3(256), 0, s, t->hi.as<uint4>(), t->zhi.as<uint4>(), 1u << t->hi_bits);
    } else {
        hipLaunchKernelGGL((ntt_tables_lazy_kernel<F, false>), dim3(((1u << t->lo_bits) + 255) / 256), dim3(256), 0, s, t->lo.as<uint4>(), t->zlo.as<uint4>(), 1u << t->lo_bits);
        hipLaunchKernelGGL((ntt_tables_lazy_kernel<F, false>), dim3(((1u << t->hi_bits) + 255) / 256), dim3(256), 0, s, t->hi.as<uint4>(), t->zhi.as<uint4>(), 1u << t->hi_bits);
    }
    // direct inter-pass tables for the lazy passes (pass p >= 1 reads Ns * R entries, coalesced): up to 1 GiB per pass
    static const int direct_on = getenv("TRH_NTT_DIRECT") ? atoi(getenv("TRH_NTT_DIRECT")) : 1;
    int sizes[8], P, tlog;
    plan_passes(log_n, sizes, &P, &tlog);
    if (direct_on && lazy_enabled() && tlog == TILE_LOG && log_n >= TILE_LOG) {
        int log_ns = sizes[0];
        for (int p = 1; p < P && rc == TRH_OK; ++p) {
            const size_t entries = (size_t)1 << (log_ns + sizes[p]);
            const size_t entry_bytes = signed_enabled() ? 36 : 32;  // raw limbs for the signed passes
            if (entries * entry_bytes <= ((size_t)1 << 30) + ((size_t)1 << 27)) {
                rc = t->direct[p].ensure(entries * entry_bytes);
                if (rc == TRH_OK && signed_enabled()) {
                    const bool fold = scale && p == P - 1;
                    const u32* sw = (const u32*)t->scale;
                    hipLaunchKernelGGL((ntt_direct_table_y_kernel<F>), dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, s, t->zlo.as<uint4>(), t->zhi.as<uint4>(), t->lo_bits,
                                       t->direct[p].as<uint4>(), log_ns, sizes[p], log_n - log_ns - sizes[p], make_uint4(sw[0], sw[1], sw[2], sw[3]), make_uint4(sw[4], sw[5], sw[6], sw[7]),
                                       fold ? 1 : 0);
                    if (fold) t->scaled = true;
                }
                else if (rc == TRH_OK)
                    hipLaunchKernelGGL((ntt_direct_table_kernel<F>), dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, s, t->zlo.as<uint4>(), t->zhi.as<uint4>(), t->lo_bits,
                                       t->direct[p].as<uint4>(), log_ns, sizes[p], log_n - log_ns - sizes[p]);
            }
            log_ns += sizes[p];
        }
    }
    static const int half_on = getenv("TRH_NTT_HALF") ? atoi(getenv("TRH_NTT_HALF")) : 1;  // 0: 9-stage passes with packed in-tile twiddles (A/B)
    bool nine = false;
    for (int p = 0; p < P; ++p) nine = nine || sizes[p] == 9;
    if (rc == TRH_OK && half_on && nine && signed_enabled() && tlog == TILE_LOG && log_n >= TILE_LOG) {
        rc = t->tile9.ensure((size_t)256 * 36);
        if (rc == TRH_OK)
            hipLaunchKernelGGL((ntt_tile_table_y_kernel<F>), dim3(1), dim3(256), 0, s, t->zlo.as<uint4>(), t->zhi.as<uint4>(), t->lo_bits, t->tile9.as<uint4>(), log_n, 9);
    }
    TRH_HIP_TRY(hipGetLastError());
    TRH_HIP_TRY(hipStreamSynchronize(s));  // pw is a stack buffer
    if (rc != TRH_OK) { t->release_all(); delete t; return rc; }
    c.twiddles.push_back(t);
    *out = t;
    return TRH_OK;
}

template <class F>
int ntt_device_t(void* a_dev, uint32_t log_n, const u64 omega[4], size_t batch, hipStream_t s, const NttFusion* fu, const u64* scale) {
    if (log_n == 0 || batch == 0) return TRH_OK;
    Ctx& c = ctx();
    TwiddleEntry* t = find_tables(F::ID, (int)log_n, omega, scale);
    if (!t) TRH_TRY(build_tables<F>((int)log_n, omega, scale, s, &t));

    int sizes[8], P = 0, tlog = TILE_LOG;
    plan_passes((int)log_n, sizes, &P, &tlog);
    const size_t N = (size_t)1 << log_n;
    uint4* a = (uint4*)a_dev;
    bool all_lazy = lazy_enabled() && tlog == TILE_LOG && (int)log_n >= TILE_LOG && P >= 2;
    for (int p = 0; p < P; ++p) all_lazy = all_lazy && sizes[p] >= 2 && sizes[p] <= MAX_PASS_LOG;
    if (scale && !(all_lazy && signed_enabled() && t->scaled)) { set_error("ntt: this size cannot take a factor in its tables (ntt_can_fold_scale)"); return TRH_EINVAL; }
    if (all_lazy && signed_enabled()) {
        // signed-domain passes: caller's words -> raw nine-limb scratch -> ... -> caller's words (canonical)
        const size_t max_tmp = (size_t)2 << 30;
        size_t chunk = max_tmp / (N * 72);
        if (chunk < 1) chunk = 1;
        if (fu && fu->blocks) chunk = chunk < fu->blocks ? fu->blocks : chunk - chunk % fu->blocks;  // whole polynomials per launch: block = index % blocks
        if (chunk > batch) chunk = batch;
        TRH_TRY(c.ntt_tmp.ensure(2 * chunk * N * 36));
        char* raw[2] = {(char*)c.ntt_tmp.p, (char*)c.ntt_tmp.p + chunk * N * 36};
        for (size_t b0 = 0; b0 < batch; b0 += chunk) {
            const size_t nb = (b0 + chunk <= batch) ? chunk : batch - b0;
            uint4* base = a + b0 * N * 2;
            int log_ns = 0;
            for (int p = 0; p < P; ++p) {
                const int sp = sizes[p];
                static const int grid_knob = getenv("TRH_NTT_GRID") ? atoi(getenv("TRH_NTT_GRID")) : -1;  // 0 tile-major, 1 batch-major, default: automatic
                const int batch_major = grid_knob >= 0 ? (grid_knob && (N >> TILE_LOG) <= 65535) : (nb >= 8 && p > 0 && (N >> TILE_LOG) <= 65535);
                const dim3 grid = batch_major ? dim3((unsigned)nb, (unsigned)(N >> TILE_LOG)) : dim3((unsigned)(N >> TILE_LOG), (unsigned)nb);
                const uint4* direct = t->direct[p].p ? t->direct[p].as<uint4>() : nullptr;
                NttFusion kf;
                if (fu && p == 0) {
                    kf.pre = fu->pre; kf.pre_period = fu->pre_period;
                    const size_t row0 = fu->pre_blocks ? b0 / fu->blocks : b0;
                    if (fu->in_dev) { kf.in_dev = (const char*)fu->in_dev + row0 * ((size_t)32 << fu->in_log); kf.in_log = fu->in_log; }
                    kf.pre_blocks = fu->pre_blocks;
                }
                if (fu && p == P - 1) { kf.post = fu->post; kf.post_period = fu->post_period; kf.post_blocks = fu->post_blocks; }
                if (fu) kf.blocks = fu->blocks;
                const bool fused = kf.in_dev || kf.pre || kf.post || kf.pre_blocks || kf.post_blocks;
                const uint4* src = p == 0 ? base : (const uint4*)raw[(p - 1) & 1];
                uint4* dst = p == P - 1 ? base : (uint4*)raw[p & 1];
                const size_t ldz = ((size_t)36 << TILE_LOG) + ((size_t)32 << (sp - 1)), ldl = ((size_t)36 << TILE_LOG) + ((size_t)36 << (sp - 1));
                const size_t ldh = ((size_t)36 << TILE_LOG) + ((size_t)36 << (sp - 2));
                const uint4* tile_tab = (sp == 9 && t->tile9.p) ? t->tile9.as<uint4>() : nullptr;
                // persistent software-pipelined form (ntt_passp_kernel) for the shapes it takes: TRH_NTT_PIPE=1.  Off by default -- measured
                // (profiles/r05_ntt_pipeline_ab.txt): 2^22 0.437 vs 0.439 ms, 2^24 +2 %, 320 x 2^18 +10 %, coset blocks +8 %; the waves wait less
                // (SQ_WAIT_ANY -20 %) and the pass takes as long: what bounds it is VALU issue, not the load / store phases
                static const int pipe_knob = getenv("TRH_NTT_PIPE") ? atoi(getenv("TRH_NTT_PIPE")) : 0;
                static const int pipe_wg = getenv("TRH_NTT_PIPE_WG") ? atoi(getenv("TRH_NTT_PIPE_WG")) : 2;  // resident workgroups per CU
                const bool pipe_ok = pipe_knob && log_n <= 26 && sp >= 4 && (sp <= 8 || tile_tab) && !kf.pre && !kf.post && (!kf.in_dev || kf.in_log == log_n) && (p == 0 || direct) &&
                                     (N >> TILE_LOG) * nb < ((size_t)1 << 31);
                if (pipe_ok) {
                    NttPipeArgs pa;
                    pa.in = (p == 0 && kf.in_dev) ? (const uint4*)kf.in_dev : src;
                    pa.out = dst;
                    pa.log_n = (int)log_n; pa.s = sp; pa.log_ns = log_ns; pa.last = (int)(p == P - 1); pa.words_in = (int)(p == 0); pa.raw_out = (int)(p < P - 1);
                    pa.fac = p > 0 ? direct : (const uint4*)kf.pre_blocks;
                    pa.fac_M = p > 0 ? (unsigned long long)1 << (log_ns + sp) : (unsigned long long)kf.blocks << log_n;
                    pa.fac_blocks = (p == 0 && kf.pre_blocks) ? kf.blocks : 0u;
                    pa.in_div = (p == 0 && kf.pre_blocks) ? kf.blocks : 1u;
                    pa.post_blocks = (const uint4*)kf.post_blocks; pa.blocks = kf.blocks;
                    pa.nb = (u32)nb; pa.batch_major = batch_major;
                    pa.tile_tab = tile_tab; pa.z_lo = t->zlo.as<uint4>(); pa.z_hi = t->zhi.as<uint4>(); pa.lo_bits = t->lo_bits;
                    if (!c.cu_count) TRH_HIP_TRY(hipDeviceGetAttribute(&c.cu_count, hipDeviceAttributeMultiprocessorCount, c.device));
                    const size_t total = (N >> TILE_LOG) * nb, resident = (size_t)c.cu_count * (size_t)(pipe_wg > 0 ? pipe_wg : 2);
                    const dim3 pgrid((unsigned)(total < resident ? total : resident));
                    const size_t lds = sp <= 8 ? ldl : ldh;
                    const int mode = (p == 0 ? 1 : 0) | (pa.fac ? 2 : 0) | ((p == P - 1 && kf.post_blocks) ? 4 : 0) | (p < P - 1 ? 8 : 0);
#define TRH_LAUNCH_PASSP(TWM, MODE) hipLaunchKernelGGL((ntt_passp_kernel<F, TWM, MODE>), pgrid, dim3(TILE >> 2), lds, s, pa)
#define TRH_LAUNCH_PASSP_MODES(TWM)                                                                                                                      \
    do {                                                                                                                                                 \
        if (mode == 9) TRH_LAUNCH_PASSP(TWM, 9); else if (mode == 11) TRH_LAUNCH_PASSP(TWM, 11); else if (mode == 10) TRH_LAUNCH_PASSP(TWM, 10);          \
        else if (mode == 2) TRH_LAUNCH_PASSP(TWM, 2); else TRH_LAUNCH_PASSP(TWM, 6);                                                                     \
    } while (0)
                    if (sp <= 8) TRH_LAUNCH_PASSP_MODES(1); else TRH_LAUNCH_PASSP_MODES(2);
#undef TRH_LAUNCH_PASSP_MODES
#undef TRH_LAUNCH_PASSP
                    log_ns += sp;
                    continue;
                }
#define TRH_LAUNCH_PASSY(TWM, FUSE, LDS)                                                                                                       \
    hipLaunchKernelGGL((ntt_passy_kernel<F, 2, TILE_LOG, TWM, FUSE>), grid, dim3(TILE >> 2), LDS, s, src, dst, (int)log_n, sp, log_ns, t->zlo.as<uint4>(), \
                       t->zhi.as<uint4>(), t->lo_bits, (int)(p == P - 1), (int)(p > 0), (int)(p < P - 1), direct, kf, batch_major, tile_tab)
                if (sp <= 8 && !fused) TRH_LAUNCH_PASSY(1, false, ldl);
                else if (sp <= 8) TRH_LAUNCH_PASSY(1, true, ldl);
                else if (tile_tab && !fused) TRH_LAUNCH_PASSY(2, false, ldh);
                else if (tile_tab) TRH_LAUNCH_PASSY(2, true, ldh);
                else if (!fused) TRH_LAUNCH_PASSY(0, false, ldz);
                else TRH_LAUNCH_PASSY(0, true, ldz);
#undef TRH_LAUNCH_PASSY
                log_ns += sp;
            }
        }
        TRH_HIP_TRY(hipGetLastError());
        return TRH_OK;
    }
    uint4* tmp = nullptr;
    size_t chunk = batch;
    if (P > 1) {
        const size_t max_tmp = (size_t)2 << 30;  // cap the scratch at 2 GiB per call
        chunk = max_tmp / (N * 32);
        if (chunk < 1) chunk = 1;
        if (chunk > batch) chunk = batch;
        TRH_TRY(c.ntt_tmp.ensure(chunk * N * 32));
        tmp = c.ntt_tmp.as<uint4>();
    }
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = (b0 + chunk <= batch) ? chunk : batch - b0;
        uint4* base = a + b0 * N * 2;
        uint4* src = base;
        uint4* dst = tmp;
        int log_ns = 0;
        for (int p = 0; p < P; ++p) {
            const int sp = sizes[p];
            const bool in_place = (P == 1) || ((P & 1) && p == P - 1);
            uint4* o = in_place ? src : dst;
            const int tile_log = (int)log_n < tlog ? (int)log_n : tlog;
            const size_t tiles = N >> tile_log;
            const size_t lds = ((size_t)32 << tile_log) + ((size_t)32 << (sp - 1 > 0 ? sp - 1 : 0));
            const dim3 grid((unsigned)tiles, (unsigned)nb);
            const bool lazy = lazy_enabled();
            const uint4* direct = t->direct[p].p ? t->direct[p].as<uint4>() : nullptr;
            NttFusion kf;  // what this pass fuses: input side on pass 0, output side on the last pass
            if (fu && p == 0) {
                kf.pre = fu->pre; kf.pre_period = fu->pre_period;
                if (fu->in_dev) { kf.in_dev = (const char*)fu->in_dev + b0 * ((size_t)32 << fu->in_log); kf.in_log = fu->in_log; }
            }
            if (fu && p == P - 1) { kf.post = fu->post; kf.post_period = fu->post_period; }
            const size_t ldz = ((size_t)36 << TILE_LOG) + ((size_t)32 << (sp - 1));
            const bool fused = kf.in_dev || kf.pre || kf.post;
            // (the tables are in the signed 2^261 form whenever the signed passes are on: the unsigned lazy kernel must not read them --
            //  a mixed TRH_NTT_PLAN such as "10,4" then runs its passes canonically)
            const bool lazy_pass = lazy && !signed_enabled() && tlog == TILE_LOG && (int)log_n >= TILE_LOG && sp >= 2 && sp <= MAX_PASS_LOG;
            const size_t ldl = ((size_t)36 << TILE_LOG) + ((size_t)36 << (sp - 1));
#define TRH_LAUNCH_PASSZ(TWL, FUSE, LDS)                                                                                                       \
    hipLaunchKernelGGL((ntt_passz_kernel<F, 2, TILE_LOG, TWL, FUSE>), grid, dim3(TILE >> 2), LDS, s, src, o, (int)log_n, sp, log_ns, t->zlo.as<uint4>(), \
                       t->zhi.as<uint4>(), t->lo_bits, (int)(p == P - 1), direct, kf)
            if (lazy_pass && sp <= 8 && !fused) TRH_LAUNCH_PASSZ(true, false, ldl);
            else if (lazy_pass && sp <= 8) TRH_LAUNCH_PASSZ(true, true, ldl);
            else if (lazy_pass && !fused) TRH_LAUNCH_PASSZ(false, false, ldz);
            else if (lazy_pass) TRH_LAUNCH_PASSZ(false, true, ldz);
#undef TRH_LAUNCH_PASSZ
            else if ((int)log_n >= TILE_LOG && sp >= 2)
                hipLaunchKernelGGL((ntt_passg_kernel<F, 2, TILE_LOG>), grid, dim3(TILE >> 2), lds, s, src, o, (int)log_n, sp, log_ns, t->lo.as<uint4>(), t->hi.as<uint4>(), t->lo_bits);
            else
                hipLaunchKernelGGL((ntt_pass_kernel<F>), grid, dim3(NTT_THREADS), lds, s, src, o, (int)log_n, sp, log_ns, t->lo.as<uint4>(), t->hi.as<uint4>(), t->lo_bits);
            if (!in_place) { uint4* x = src; src = dst; dst = x; }
            log_ns += sp;
        }
    }
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

}  // namespace

bool ntt_can_fuse(uint32_t log_n) {
    int sizes[8], P, tlog;
    plan_passes((int)log_n, sizes, &P, &tlog);
    if (!lazy_enabled() || tlog != TILE_LOG || (int)log_n < TILE_LOG) return false;
    for (int p = 0; p < P; ++p) if (sizes[p] < 2 || sizes[p] > MAX_PASS_LOG) return false;  // every pass must be a lazy one
    static const int fuse = getenv("TRH_NTT_FUSE") ? atoi(getenv("TRH_NTT_FUSE")) : 1;
    return fuse != 0;
}

bool ntt_can_fold_scale(uint32_t log_n) {
    static const int direct_on = getenv("TRH_NTT_DIRECT") ? atoi(getenv("TRH_NTT_DIRECT")) : 1;
    static const int fold_on = getenv("TRH_NTT_FOLD_SCALE") ? atoi(getenv("TRH_NTT_FOLD_SCALE")) : 1;  // 0: the factor as a separate multiplication on the last store (A/B)
    int sizes[8], P, tlog;
    plan_passes((int)log_n, sizes, &P, &tlog);
    return fold_on && direct_on && ntt_can_fuse(log_n) && signed_enabled() && P >= 2 && ((size_t)36 << log_n) <= ((size_t)1 << 30) + ((size_t)1 << 27);
}

int ntt_device(int field, void* a_dev, uint32_t log_n, const u64 omega[4], size_t batch, hipStream_t s, const NttFusion* fu, const u64* scale) {
    if (log_n > 27) { set_error("ntt: log_n %u > 27 unsupported", log_n); return TRH_EINVAL; }
    if (fu && !ntt_can_fuse(log_n)) { set_error("ntt: fused pointwise steps need the lazy passes (log_n >= %d)", TILE_LOG); return TRH_EINVAL; }
    if (!(ctx().attr_done & ATTR_NTT)) {  // per device
        const int max_lds = (32 << TILE_LOG) + (32 << (TILE_LOG - 1));
        TRH_HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel<FpParams>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        TRH_HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel<FqParams>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        TRH_HIP_TRY(hipFuncSetAttribute((const void*)ntt_passg_kernel<FpParams, 2, TILE_LOG>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        TRH_HIP_TRY(hipFuncSetAttribute((const void*)ntt_passg_kernel<FqParams, 2, TILE_LOG>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        const int z_lds = (36 << TILE_LOG) + (32 << (MAX_PASS_LOG - 1));  // 80 KiB: two workgroups per CU
#define TRH_PASSZ_ATTR(FIELD, TWL, FUSE) TRH_HIP_TRY(hipFuncSetAttribute((const void*)ntt_passz_kernel<FIELD, 2, TILE_LOG, TWL, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, z_lds))
        TRH_PASSZ_ATTR(FpParams, true, false); TRH_PASSZ_ATTR(FpParams, true, true); TRH_PASSZ_ATTR(FpParams, false, false); TRH_PASSZ_ATTR(FpParams, false, true);
        TRH_PASSZ_ATTR(FqParams, true, false); TRH_PASSZ_ATTR(FqParams, true, true); TRH_PASSZ_ATTR(FqParams, false, false); TRH_PASSZ_ATTR(FqParams, false, true);
#undef TRH_PASSZ_ATTR
#define TRH_PASSY_ATTR(FIELD, TWM, FUSE) TRH_HIP_TRY(hipFuncSetAttribute((const void*)ntt_passy_kernel<FIELD, 2, TILE_LOG, TWM, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, z_lds))
        TRH_PASSY_ATTR(FpParams, 1, false); TRH_PASSY_ATTR(FpParams, 1, true); TRH_PASSY_ATTR(FpParams, 0, false); TRH_PASSY_ATTR(FpParams, 0, true);
        TRH_PASSY_ATTR(FqParams, 1, false); TRH_PASSY_ATTR(FqParams, 1, true); TRH_PASSY_ATTR(FqParams, 0, false); TRH_PASSY_ATTR(FqParams, 0, true);
        TRH_PASSY_ATTR(FpParams, 2, false); TRH_PASSY_ATTR(FpParams, 2, true); TRH_PASSY_ATTR(FqParams, 2, false); TRH_PASSY_ATTR(FqParams, 2, true);
#undef TRH_PASSY_ATTR
#define TRH_PASSP_ATTR(FIELD, TWM, MODE) TRH_HIP_TRY(hipFuncSetAttribute((const void*)ntt_passp_kernel<FIELD, TWM, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, z_lds))
#define TRH_PASSP_ATTRS(FIELD, TWM) TRH_PASSP_ATTR(FIELD, TWM, 9); TRH_PASSP_ATTR(FIELD, TWM, 11); TRH_PASSP_ATTR(FIELD, TWM, 10); TRH_PASSP_ATTR(FIELD, TWM, 2); TRH_PASSP_ATTR(FIELD, TWM, 6)
        TRH_PASSP_ATTRS(FpParams, 1); TRH_PASSP_ATTRS(FpParams, 2); TRH_PASSP_ATTRS(FqParams, 1); TRH_PASSP_ATTRS(FqParams, 2);
#undef TRH_PASSP_ATTRS
#undef TRH_PASSP_ATTR
        ctx().attr_done |= ATTR_NTT;
    }
    if (field == TRH_FP) return ntt_device_t<FpParams>(a_dev, log_n, omega, batch, s, fu, scale);
    return ntt_device_t<FqParams>(a_dev, log_n, omega, batch, s, fu, scale);
}

int field_scale_periodic(int field, void* a_dev, size_t rows, size_t row_len, size_t active_len, const void* factors_dev, u32 period, hipStream_t s) {
    const size_t n = rows * active_len;
    if (!n) return TRH_OK;
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (field == TRH_FP) hipLaunchKernelGGL((field_scale_periodic_kernel<FpParams>), dim3(gb), dim3(256), 0, s, (uint4*)a_dev, rows, row_len, active_len, (const uint4*)factors_dev, period);
    else hipLaunchKernelGGL((field_scale_periodic_kernel<FqParams>), dim3(gb), dim3(256), 0, s, (uint4*)a_dev, rows, row_len, active_len, (const uint4*)factors_dev, period);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

size_t ntt_block_table_bytes(uint32_t blocks, uint32_t log_n) { return ((size_t)blocks << log_n) * 36; }

int ntt_block_table_build(int field, void* table_dev, uint32_t blocks, uint32_t log_n, const u64 g[4], const u64 w[4], const u64 scale[4], hipStream_t s) {
    if (blocks == 0 || blocks > 64 || log_n > 27) { set_error("ntt_block_table: bad shape"); return TRH_EINVAL; }
    Ctx& c = ctx();
    TRH_TRY(c.factors.ensure(16 * 64 * 32));
    char* slot = (char*)c.factors.p + (size_t)(c.factor_slot++ & 15) * 64 * 32;
    u64 h[12];
    memcpy(h, g, 32); memcpy(h + 4, w, 32); memcpy(h + 8, scale, 32);
    TRH_HIP_TRY(hipMemcpyAsync(slot, h, 96, hipMemcpyHostToDevice, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));  // h is a stack buffer
    const size_t M = (size_t)blocks << log_n;
    const unsigned gb = (unsigned)((M + 255) / 256);
    if (field == TRH_FP) hipLaunchKernelGGL((ntt_block_table_kernel<FpParams>), dim3(gb), dim3(256), 0, s, (uint4*)table_dev, blocks, (int)log_n, (const uint4*)slot);
    else hipLaunchKernelGGL((ntt_block_table_kernel<FqParams>), dim3(gb), dim3(256), 0, s, (uint4*)table_dev, blocks, (int)log_n, (const uint4*)slot);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

int ntt_block_scale(int field, const void* in_dev, void* out_dev, size_t transforms, uint32_t blocks, uint32_t log_n, const void* table_dev, bool in_per_block, hipStream_t s) {
    const size_t total = transforms << log_n;
    if (!total) return TRH_OK;
    const unsigned gb = (unsigned)((total + 255) / 256);
    if (field == TRH_FP) hipLaunchKernelGGL((ntt_block_scale_kernel<FpParams>), dim3(gb), dim3(256), 0, s, (const uint4*)in_dev, (uint4*)out_dev, transforms, blocks, (int)log_n, (const uint4*)table_dev, in_per_block ? 1 : 0);
    else hipLaunchKernelGGL((ntt_block_scale_kernel<FqParams>), dim3(gb), dim3(256), 0, s, (const uint4*)in_dev, (uint4*)out_dev, transforms, blocks, (int)log_n, (const uint4*)table_dev, in_per_block ? 1 : 0);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

// log2 of the factor between the lazy domain of the passes and the memory format's Montgomery radix 2^256: the pointwise factors
// fused into the passes (EvaluationDomain, domain.hip) are multiplied by 2^this
int ntt_lazy_shift() { return signed_enabled() ? 5 : 14; }

void ntt_release_tables() {
    Ctx& c = ctx();
    for (TwiddleEntry* t : c.twiddles) { t->release_all(); delete t; }
    c.twiddles.clear();
    c.ntt_tmp.release();
}

}  // namespace trh
