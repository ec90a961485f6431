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
// the same tables in the signed lazy domain's Montgomery form (x 2^261), non-negative, below m (1 + 2^-7)
template <class F>
__global__ void __launch_bounds__(256) ntt_tables_lazy_kernel(const uint4* __restrict__ t, uint4* __restrict__ z, u32 cnt) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    u32 w[8];
    fy_store(fy_from_fe(load_fe<F>(t + 2 * (size_t)i)), w);
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
    // (a 4096-element tile -- two passes for 2^19..2^22, all 160 KiB of LDS, one workgroup per CU -- measured equal: removed.  Round 6: 2^22 as
    //  two 11-stage passes on the 2048-element tile, pass 0 reading 64-KiB-strided columns: 0.83 ms against 0.47 for 8 + 7 + 7, the fabric
    //  fetches 4.8 x the bytes -- profiles/r06_ntt_11_11_ab.txt.)
    int P = 0, tlog = TILE_LOG;
    if (log_n <= TILE_LOG) { sizes[P++] = log_n; }
    else {
        P = (log_n + MAX_PASS_LOG - 1) / MAX_PASS_LOG;
        if (P < 2) P = 2;
        int rem = log_n;
        for (int p = 0; p < P; ++p) { sizes[p] = (rem + (P - p) - 1) / (P - p); rem -= sizes[p]; }
    }
    *n_passes = P; *tile_log = tlog;
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
    hipLaunchKernelGGL((ntt_tables_lazy_kernel<F>), dim3(((1u << t->lo_bits) + 255) / 256), dim3(256), 0, s, t->lo.as<uint4>(), t->zlo.as<uint4>(), 1u << t->lo_bits);
    hipLaunchKernelGGL((ntt_tables_lazy_kernel<F>), dim3(((1u << t->hi_bits) + 255) / 256), dim3(256), 0, s, t->hi.as<uint4>(), t->zhi.as<uint4>(), 1u << t->hi_bits);
    // direct inter-pass tables for the lazy passes (pass p >= 1 reads Ns * R entries, coalesced): up to 1 GiB per pass
    int sizes[8], P, tlog;
    plan_passes(log_n, sizes, &P, &tlog);
    if (tlog == TILE_LOG && log_n >= TILE_LOG) {
        int log_ns = sizes[0];
        for (int p = 1; p < P && rc == TRH_OK; ++p) {
            const size_t entries = (size_t)1 << (log_ns + sizes[p]);
            const size_t entry_bytes = 36;  // raw limbs
            if (entries * entry_bytes <= ((size_t)1 << 30) + ((size_t)1 << 27)) {
                rc = t->direct[p].ensure(entries * entry_bytes);
                if (rc == TRH_OK) {
                    const bool fold = scale && p == P - 1;
                    const u32* sw = (const u32*)t->scale;
                    hipLaunchKernelGGL((ntt_direct_table_y_kernel<F>), dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, s, t->zlo.as<uint4>(), t->zhi.as<uint4>(), t->lo_bits,
                                       t->direct[p].as<uint4>(), log_ns, sizes[p], log_n - log_ns - sizes[p], make_uint4(sw[0], sw[1], sw[2], sw[3]), make_uint4(sw[4], sw[5], sw[6], sw[7]),
                                       fold ? 1 : 0);
                    if (fold) t->scaled = true;
                }
            }
            log_ns += sizes[p];
        }
    }
    bool nine = false;
    for (int p = 0; p < P; ++p) nine = nine || sizes[p] == 9;
    if (rc == TRH_OK && nine && tlog == TILE_LOG && log_n >= TILE_LOG) {
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
    bool all_lazy = tlog == TILE_LOG && (int)log_n >= TILE_LOG && P >= 2;
    for (int p = 0; p < P; ++p) all_lazy = all_lazy && sizes[p] >= 2 && sizes[p] <= MAX_PASS_LOG;
    if (scale && !(all_lazy && t->scaled)) { set_error("ntt: this size cannot take a factor in its tables (ntt_can_fold_scale)"); return TRH_EINVAL; }
    if (all_lazy) {
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
                const int batch_major = nb >= 8 && p > 0 && (N >> TILE_LOG) <= 65535;  // consecutive workgroups run the same tile of consecutive transforms: the inter-pass table's rows are re-read from the L2
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
                const size_t ldl = ((size_t)36 << TILE_LOG) + ((size_t)36 << (sp - 1));
                const size_t ldh = ((size_t)36 << TILE_LOG) + ((size_t)36 << (sp - 2));
                const uint4* tile_tab = (sp == 9 && t->tile9.p) ? t->tile9.as<uint4>() : nullptr;
#define TRH_LAUNCH_PASSY(TWM, FUSE, LDS)                                                                                                       \
    hipLaunchKernelGGL((ntt_passy_kernel<F, 2, TILE_LOG, TWM, FUSE>), grid, dim3(TILE >> 2), LDS, s, src, dst, (int)log_n, sp, log_ns, t->zlo.as<uint4>(), \
                       t->zhi.as<uint4>(), t->lo_bits, (int)(p == P - 1), (int)(p > 0), (int)(p < P - 1), direct, kf, batch_major, tile_tab)
                if (sp <= 8 && !fused) TRH_LAUNCH_PASSY(1, false, ldl);
                else if (sp <= 8) TRH_LAUNCH_PASSY(1, true, ldl);
                else if (!tile_tab) { set_error("ntt: 9-stage pass without its tile table"); return TRH_EINVAL; }  // (build_tables makes one whenever a pass has nine stages)
                else if (!fused) TRH_LAUNCH_PASSY(2, false, ldh);
                else TRH_LAUNCH_PASSY(2, true, ldh);
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
            // (sizes up to one tile: canonical passes; ntt_can_fuse is false here, so there is nothing fused to apply)
            if ((int)log_n >= TILE_LOG && sp >= 2)
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
    if (tlog != TILE_LOG || (int)log_n < TILE_LOG) return false;
    for (int p = 0; p < P; ++p) if (sizes[p] < 2 || sizes[p] > MAX_PASS_LOG) return false;  // every pass must be a lazy one
    return true;
}

bool ntt_can_fold_scale(uint32_t log_n) {
    int sizes[8], P, tlog;
    plan_passes((int)log_n, sizes, &P, &tlog);
    return ntt_can_fuse(log_n) && P >= 2 && ((size_t)36 << log_n) <= ((size_t)1 << 30) + ((size_t)1 << 27);
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
#define TRH_PASSY_ATTR(FIELD, TWM, FUSE) TRH_HIP_TRY(hipFuncSetAttribute((const void*)ntt_passy_kernel<FIELD, 2, TILE_LOG, TWM, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, z_lds))
        TRH_PASSY_ATTR(FpParams, 1, false); TRH_PASSY_ATTR(FpParams, 1, true);
        TRH_PASSY_ATTR(FqParams, 1, false); TRH_PASSY_ATTR(FqParams, 1, true);
        TRH_PASSY_ATTR(FpParams, 2, false); TRH_PASSY_ATTR(FpParams, 2, true); TRH_PASSY_ATTR(FqParams, 2, false); TRH_PASSY_ATTR(FqParams, 2, true);
#undef TRH_PASSY_ATTR
        ctx().attr_done |= ATTR_NTT;
    }
    if (field == TRH_FP) return ntt_device_t<FpParams>(a_dev, log_n, omega, batch, s, fu, scale);
    return ntt_device_t<FqParams>(a_dev, log_n, omega, batch, s, fu, scale);
}

int ntt_prepare(int field, uint32_t log_n, const u64 omega[4], const u64* scale, size_t batch, uint32_t blocks, hipStream_t s) {
    if (log_n == 0 || log_n > 27 || batch == 0) return TRH_OK;
    Ctx& c = ctx();
    TwiddleEntry* t = find_tables(field, (int)log_n, omega, scale);
    if (!t) TRH_TRY(field == TRH_FP ? build_tables<FpParams>((int)log_n, omega, scale, s, &t) : build_tables<FqParams>((int)log_n, omega, scale, s, &t));
    int sizes[8], P = 0, tlog = TILE_LOG;
    plan_passes((int)log_n, sizes, &P, &tlog);
    const size_t N = (size_t)1 << log_n;
    if (ntt_can_fuse(log_n) && P >= 2) {  // the signed passes' raw scratch: the chunking of ntt_device_t
        size_t chunk = ((size_t)2 << 30) / (N * 72);
        if (chunk < 1) chunk = 1;
        if (blocks) chunk = chunk < blocks ? blocks : chunk - chunk % blocks;
        if (chunk > batch) chunk = batch;
        return c.ntt_tmp.ensure(2 * chunk * N * 36);
    }
    if (P > 1) {
        size_t chunk = ((size_t)2 << 30) / (N * 32);
        if (chunk < 1) chunk = 1;
        if (chunk > batch) chunk = batch;
        return c.ntt_tmp.ensure(chunk * N * 32);
    }
    return TRH_OK;
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
int ntt_lazy_shift() { return 5; }

void ntt_release_tables() {
    Ctx& c = ctx();
    for (TwiddleEntry* t : c.twiddles) { t->release_all(); delete t; }
    c.twiddles.clear();
    c.ntt_tmp.release();
}

}  // namespace trh
