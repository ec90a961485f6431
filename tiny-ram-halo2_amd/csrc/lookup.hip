// The permuted columns of halo2_proofs 0.2.0's lookup argument (plonk/lookup/prover.rs `permute_expression_pair`, reached
// from create_proof -- /root/reference/src/test_utils.rs:41-49; the reference's circuit has 31 lookups,
// src/circuits/even_bits.rs:158-170, aux/out_table.rs:33-74, shift.rs:142-165, tables/prog.rs:170-192):
//
//   A' = the first `usable_rows` input values sorted (Ord of the field = order of the canonical integers)
//   S'[row] = A'[row] where A'[row] is the first of its run; one instance of that value leaves the table multiset;
//   the other rows receive the left-over table values in ascending order, the LAST repeated row first
//   (the Rust code pops the rows of a Vec and walks a BTreeMap);  an input value missing from the table is an error.
//
// All lookups of a proof go through ONE set of launches (blockIdx.y = column: the `batch` inputs, then the `batch` tables) and
// one host synchronisation (the error / tie flags).  Sorting 256-bit keys: field elements are either small (range tables: only
// the low limb varies) or spread over the whole field (compressed expressions: two different values practically never share
// their top limb), so a stable LSD radix sort of (most significant varying 64-bit limb, row) pairs is almost always the complete
// order -- checked on the device; a column with a tie between different values is redone with the general form, the same sort
// over every varying limb, least significant first.  The radix sort is this file's own (4-bit digits, per-thread digit counts
// in LDS for a stable block-local rank, one scan of the (digit, block) histogram per pass); a pass whose digit is constant over
// a column -- all but four of the sixteen for a 16-bit range table -- returns at once, the ping-pong side of every column being
// tracked on the device.  Not a hot kernel of the path (18 ms of a 390 ms proof before, a few ms now).
#include <string.h>

#include <cstring>
#include <vector>

#include "ctx.h"

namespace trh {
namespace {

constexpr int TILE = 2048;      // elements per workgroup of the sort / scan kernels
constexpr int THREADS = 256;
constexpr int ITEMS = TILE / THREADS;
constexpr int RADIX_BITS = 4, RADIX = 1 << RADIX_BITS, PASSES = 64 / RADIX_BITS;
constexpr int FAST_KEY_BITS = 48;  // varying bits the fast path sorts by (see select_limb_kernel)

template <class F>
__device__ __forceinline__ Fe<F> ldf(const uint4* p) {
    uint4 a = p[0], b = p[1];
    return fe_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}

// column c of the batch: c < batch the input of lookup c, else the table of lookup c - batch
__device__ __forceinline__ const uint4* column_ptr(const uint4* in, const uint4* tab, size_t stride, u32 batch, u32 c) {
    return c < batch ? in + 2 * (size_t)c * stride : tab + 2 * (size_t)(c - batch) * stride;
}

// canonical limbs of a[i] (Montgomery -> integer), one u64 plane per limb; perm = identity
template <class F>
__global__ void __launch_bounds__(256) canon_planes_kernel(const uint4* __restrict__ in, const uint4* __restrict__ tab, size_t stride, u32 batch, size_t n,
                                                           u64* __restrict__ planes /* cols x 4 x n */, u32* __restrict__ perm /* cols x n */) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 c = blockIdx.y;
    if (i >= n) return;
    u32 w[8];
    fe_store(fe_from_mont(ldf<F>(column_ptr(in, tab, stride, batch, c) + 2 * i)), w);
    u64* pl = planes + (size_t)c * 4 * n;
    for (int k = 0; k < 4; ++k) pl[(size_t)k * n + i] = (u64)w[2 * k] | ((u64)w[2 * k + 1] << 32);
    perm[(size_t)c * n + i] = (u32)i;
}
// varies[c][k] = OR over the column of (limb k XOR limb k of row 0): the bits in which the column is not constant.  One atomic per
// workgroup and limb, and only when it would add a bit (every wave hitting the same four words cost 2.9 ms for 62 columns of 2^18)
__global__ void __launch_bounds__(256) plane_varies_kernel(const u64* __restrict__ planes, size_t n, u64* __restrict__ varies) {
    __shared__ unsigned long long acc[4];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 c = blockIdx.y;
    const u64* pl = planes + (size_t)c * 4 * n;
    if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        u64 x = i < n ? pl[(size_t)k * n + i] ^ pl[(size_t)k * n] : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x |= __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicOr(&acc[k], (unsigned long long)x);
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long x = acc[threadIdx.x];
        volatile const u64* cur = &varies[c * 4 + threadIdx.x];
        if (x & ~*cur) atomicOr((unsigned long long*)&varies[c * 4 + threadIdx.x], x);
    }
}
// which limb a column is sorted by in this step, and the bits of it that vary.  step < 0: the most significant varying limb (the fast
// path); step = 0 .. 3: limb `step` of the general path (key_limb = -1 when that limb is constant: the step is an identity)
__global__ void select_limb_kernel(const u64* __restrict__ varies, u32 cols, int step, int* __restrict__ key_limb, u64* __restrict__ key_mask, u32* __restrict__ side) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    int k = -1;
    if (step < 0) { for (int j = 3; j >= 0; --j) if (varies[c * 4 + j]) { k = j; break; } }
    else if (varies[c * 4 + step]) k = step;
    key_limb[c] = k;
    u64 mask = k >= 0 ? varies[c * 4 + k] : 0ull;
    if (step < 0) {
        // the fast path only has to SEPARATE the column's values: FAST_KEY_BITS varying bits from the top do that for 2^18 spread-out
        // values with probability 1 - 2^-13 per column, and a tie between different values is detected (tie_check_kernel compares
        // under this mask) and redone in the general form anyway -- so the low digits beyond them are not sorted by: 12 passes, not 16
        for (int d = 0; d < PASSES && __popcll(mask >> (RADIX_BITS * (d + 1))) >= FAST_KEY_BITS; ++d) mask &= ~((u64)(RADIX - 1) << (RADIX_BITS * d));
    }
    key_mask[c] = mask;
    (void)side;
}
// keys[side[c]][c][i] = limb key_limb[c] of the element the current order has at position i
__global__ void __launch_bounds__(256) gather_keys_kernel(const u64* __restrict__ planes, size_t n, u32 cols, const int* __restrict__ key_limb, const u32* __restrict__ side,
                                                          const u32* __restrict__ perm0, const u32* __restrict__ perm1, u64* __restrict__ keys0, u64* __restrict__ keys1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 c = blockIdx.y;
    const int k = key_limb[c];
    if (i >= n || k < 0) return;
    const u32 sd = side[c];
    const u32* perm = (sd ? perm1 : perm0) + (size_t)c * n;
    u64* keys = (sd ? keys1 : keys0) + (size_t)c * n;
    keys[i] = planes[((size_t)c * 4 + k) * n + perm[i]];
}

__device__ __forceinline__ bool pass_is_identity(u64 mask, int pass) { return ((mask >> (RADIX_BITS * pass)) & (u64)(RADIX - 1)) == 0; }

// ---- one pass of the stable LSD radix sort of (key, row) pairs, every column of the batch at once ---------------------------------
// hist[c][digit][block] = entries of the block's tile with that digit
__global__ void __launch_bounds__(THREADS) radix_hist_kernel(const u64* __restrict__ keys0, const u64* __restrict__ keys1, const u32* __restrict__ side, const u64* __restrict__ key_mask,
                                                             size_t n, int pass, u32* __restrict__ hist) {
    const u32 c = blockIdx.y, nblk = gridDim.x;
    if (pass_is_identity(key_mask[c], pass)) return;
    __shared__ u32 tot[RADIX];
    if (threadIdx.x < RADIX) tot[threadIdx.x] = 0;
    __syncthreads();
    const u64* keys = (side[c] ? keys1 : keys0) + (size_t)c * n;
    const size_t base = (size_t)blockIdx.x * TILE + (size_t)threadIdx.x * ITEMS;
    u32 cnt[RADIX];
#pragma unroll
    for (int d = 0; d < RADIX; ++d) cnt[d] = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q)
        if (base + q < n) {
            const u32 d = (u32)(keys[base + q] >> (RADIX_BITS * pass)) & (RADIX - 1);
#pragma unroll
            for (int e = 0; e < RADIX; ++e) cnt[e] += (d == (u32)e);
        }
#pragma unroll
    for (int d = 0; d < RADIX; ++d) {
        u32 v = cnt[d];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&tot[d], v);
    }
    __syncthreads();
    if (threadIdx.x < RADIX) hist[((size_t)c * RADIX + threadIdx.x) * nblk + blockIdx.x] = tot[threadIdx.x];
}
// exclusive scan of hist[c][..] in (digit, block) order -> global base of every (digit, block) run; fixes which side the scatter of
// this pass reads (sel[pass][c]) and flips the column's side for the next pass
__global__ void __launch_bounds__(THREADS) radix_scan_kernel(u32* __restrict__ hist, u32 nblk, const u64* __restrict__ key_mask, int pass, u32* __restrict__ side, u32* __restrict__ sel) {
    const u32 c = blockIdx.x;
    if (threadIdx.x == 0) sel[(size_t)pass * gridDim.x + c] = side[c];
    if (pass_is_identity(key_mask[c], pass)) return;
    __shared__ u32 part[THREADS];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    u32* h = hist + (size_t)c * RADIX * nblk;
    const u32 total = RADIX * nblk;
    for (u32 b0 = 0; b0 < total; b0 += THREADS * 8) {
        const u32 lo = b0 + threadIdx.x * 8;
        u32 v[8], sum = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { v[q] = lo + q < total ? h[lo + q] : 0u; sum += v[q]; }
        part[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 1; off < THREADS; off <<= 1) {
            const u32 t = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        u32 run = carry + part[threadIdx.x] - sum;
#pragma unroll
        for (int q = 0; q < 8; ++q) { if (lo + q < total) h[lo + q] = run; run += v[q]; }
        __syncthreads();
        if (threadIdx.x == THREADS - 1) carry += part[THREADS - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) side[c] ^= 1u;
}
// stable scatter: the tile's pairs go to base(digit, block) + rank inside the tile (threads own consecutive items, so the rank is the
// prefix of the per-thread digit counts in (digit, thread) order plus the item's position among the thread's own)
__global__ void __launch_bounds__(THREADS) radix_scatter_kernel(const u64* __restrict__ keys0, const u64* __restrict__ keys1, const u32* __restrict__ perm0, const u32* __restrict__ perm1,
                                                                u64* __restrict__ okeys0, u64* __restrict__ okeys1, u32* __restrict__ operm0, u32* __restrict__ operm1,
                                                                const u32* __restrict__ sel, const u64* __restrict__ key_mask, size_t n, int pass, const u32* __restrict__ hist) {
    const u32 c = blockIdx.y, nblk = gridDim.x;
    if (pass_is_identity(key_mask[c], pass)) return;
    __shared__ u32 cntT[RADIX * THREADS];  // [digit][thread], then its exclusive scan
    __shared__ u32 part[THREADS];
    const u32 sd = sel[(size_t)pass * gridDim.y + c];
    const u64* keys = (sd ? keys1 : keys0) + (size_t)c * n;
    const u32* perm = (sd ? perm1 : perm0) + (size_t)c * n;
    u64* okeys = (sd ? okeys0 : okeys1) + (size_t)c * n;  // the other side
    u32* operm = (sd ? operm0 : operm1) + (size_t)c * n;
    const size_t base = (size_t)blockIdx.x * TILE + (size_t)threadIdx.x * ITEMS;
    u64 k[ITEMS];
    u32 p[ITEMS], dg[ITEMS], cnt[RADIX];
#pragma unroll
    for (int d = 0; d < RADIX; ++d) cnt[d] = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        dg[q] = RADIX;  // beyond the column: no digit
        if (base + q < n) {
            k[q] = keys[base + q]; p[q] = perm[base + q];
            dg[q] = (u32)(k[q] >> (RADIX_BITS * pass)) & (RADIX - 1);
#pragma unroll
            for (int e = 0; e < RADIX; ++e) cnt[e] += (dg[q] == (u32)e);
        }
    }
#pragma unroll
    for (int d = 0; d < RADIX; ++d) cntT[d * THREADS + threadIdx.x] = cnt[d];
    __syncthreads();
    {   // exclusive scan of the RADIX * THREADS counters: thread t owns entries [16 t, 16 t + 16)
        u32 v[RADIX], sum = 0;
#pragma unroll
        for (int q = 0; q < RADIX; ++q) { v[q] = cntT[threadIdx.x * RADIX + q]; sum += v[q]; }
        part[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 1; off < THREADS; off <<= 1) {
            const u32 t = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        u32 run = part[threadIdx.x] - sum;
#pragma unroll
        for (int q = 0; q < RADIX; ++q) { cntT[threadIdx.x * RADIX + q] = run; run += v[q]; }
    }
    __syncthreads();
    // the tile's pairs are first put in sorted order in LDS (position = exclusive prefix of the counters in (digit, thread) order), then written
    // out with consecutive lanes on consecutive addresses of a digit's run: every 64-byte segment of the output receives one write instead
    // of eight partial ones from eight different instructions (the pass was bound by those L2 transactions: 260 -> ~170 us for 62 x 2^18 pairs)
    __shared__ u64 stage_k[TILE];
    __shared__ u32 stage_p[TILE];
    __shared__ u32 delta[RADIX];  // global base of the digit's run for this tile - the digit's first position inside the tile
    u32 next[RADIX];
#pragma unroll
    for (int d = 0; d < RADIX; ++d) next[d] = cntT[d * THREADS + threadIdx.x];
    if (threadIdx.x < RADIX) delta[threadIdx.x] = hist[((size_t)c * RADIX + threadIdx.x) * nblk + blockIdx.x] - cntT[threadIdx.x * THREADS];
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        if (dg[q] == RADIX) continue;
        u32 lp = 0;
#pragma unroll
        for (int e = 0; e < RADIX; ++e) if (dg[q] == (u32)e) { lp = next[e]; next[e] = lp + 1; }
        stage_k[lp] = k[q]; stage_p[lp] = p[q];
    }
    __syncthreads();
    const size_t tile_base = (size_t)blockIdx.x * TILE;
    const u32 valid = tile_base + TILE <= n ? (u32)TILE : (u32)(n - tile_base);
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        const u32 i = (u32)q * THREADS + threadIdx.x;
        if (i >= valid) break;
        const u64 key = stage_k[i];
        const u32 dst = delta[(u32)(key >> (RADIX_BITS * pass)) & (RADIX - 1)] + i;
        okeys[dst] = key; operm[dst] = stage_p[i];
    }
}

// the sorted order of column c lives on side[c] after the passes: bring it to side 0 (perm0) for the steps that follow
__global__ void __launch_bounds__(256) settle_perm_kernel(u32* __restrict__ perm0, const u32* __restrict__ perm1, u32* __restrict__ side, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 c = blockIdx.y;
    if (i < n && side[c]) perm0[(size_t)c * n + i] = perm1[(size_t)c * n + i];
}
__global__ void reset_side_kernel(u32* __restrict__ side, u32 cols) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cols) side[c] = 0;
}

// compare two elements of (possibly different) columns by their canonical 256-bit values
__device__ __forceinline__ int cmp_keys(const u64* __restrict__ pa, const u32* __restrict__ perma, size_t i, const u64* __restrict__ pb, const u32* __restrict__ permb, size_t j, size_t n) {
    const u32 ia = perma[i], ib = permb[j];
    for (int k = 3; k >= 0; --k) {
        const u64 x = pa[(size_t)k * n + ia], y = pb[(size_t)k * n + ib];
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
// after the fast path: bad[c] = 1 when two neighbours agree in the limb sorted by but are different values (their order is then unknown)
__global__ void __launch_bounds__(256) tie_check_kernel(const u64* __restrict__ planes, size_t n, const u32* __restrict__ perm, const int* __restrict__ key_limb, const u64* __restrict__ key_mask,
                                                        const u64* __restrict__ keys0, const u64* __restrict__ keys1, const u32* __restrict__ side, u32* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 c = blockIdx.y;
    const int primary = key_limb[c];
    if (i == 0 || i >= n || primary < 0) return;
    // the sorted keys lie in order on the side the last pass wrote (coalesced); the planes are only gathered for neighbours the sort did not separate
    const u64* keys = (side[c] ? keys1 : keys0) + (size_t)c * n;
    if ((keys[i - 1] ^ keys[i]) & key_mask[c]) return;  // separated by the bits sorted by
    const u64* pl = planes + (size_t)c * 4 * n;
    const u32 a = perm[(size_t)c * n + i - 1], b = perm[(size_t)c * n + i];
    for (int k = 0; k < 4; ++k)
        if (pl[(size_t)k * n + a] != pl[(size_t)k * n + b]) { bad[c] = 1u; return; }
}
// out[l][i] = column l's element at sorted position i
__global__ void __launch_bounds__(256) gather_elems_kernel(const uint4* __restrict__ in, size_t stride, const u32* __restrict__ perm, uint4* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 l = blockIdx.y;
    if (i >= n) return;
    const uint4* p = in + 2 * ((size_t)l * stride + perm[(size_t)l * n + i]);
    uint4* o = out + 2 * ((size_t)l * stride + i);
    o[0] = p[0]; o[1] = p[1];
}
// flags[l][i] = 1 when sorted input element i REPEATS its predecessor (0 at run starts); read from the gathered sorted column (equal
// Montgomery words <=> equal values; coalesced, where the planes would be two random gathers per row)
__global__ void __launch_bounds__(256) repeat_flags_kernel(const uint4* __restrict__ a_sorted, size_t stride, size_t n, u32* __restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 l = blockIdx.y;
    if (i >= n) return;
    u32 rep = 0u;
    if (i != 0) {
        const uint4* p = a_sorted + 2 * ((size_t)l * stride + i);
        const uint4 a0 = p[0], a1 = p[1], b0 = p[-2], b1 = p[-1];
        rep = (a0.x == b0.x && a0.y == b0.y && a0.z == b0.z && a0.w == b0.w && a1.x == b1.x && a1.y == b1.y && a1.z == b1.z && a1.w == b1.w) ? 1u : 0u;
    }
    flags[(size_t)l * n + i] = rep;
}
// every run start of the sorted input removes the first table instance of its value: keep[l][pos] = 0 (initialised to 1); missing -> err[l] = 1
__global__ void __launch_bounds__(256) remove_from_table_kernel(const u64* __restrict__ planes, const u32* __restrict__ perm, const u32* __restrict__ repeats, size_t n, u32 batch,
                                                                u32* __restrict__ keep, u32* __restrict__ err) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 l = blockIdx.y;
    if (i >= n || repeats[(size_t)l * n + i]) return;
    const u64 *pa = planes + (size_t)l * 4 * n, *ps = planes + (size_t)(batch + l) * 4 * n;
    const u32 *perma = perm + (size_t)l * n, *perms = perm + (size_t)(batch + l) * n;
    size_t lo = 0, hi = n;  // lower bound of the value in the sorted table
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (cmp_keys(ps, perms, mid, pa, perma, i, n) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo < n && cmp_keys(ps, perms, lo, pa, perma, i, n) == 0) keep[(size_t)l * n + lo] = 0u;  // distinct run starts hit distinct positions
    else err[l] = 1u;
}
__global__ void __launch_bounds__(256) fill_u32_kernel(u32* __restrict__ a, size_t count, u32 v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) a[i] = v;
}

// ---- exclusive scan of u32 flags, every row of the batch at once: tile scans, scan of the tile sums, add ------------------------------
__global__ void __launch_bounds__(THREADS) scan_tiles_kernel(const u32* __restrict__ in, u32* __restrict__ out, size_t n, u32* __restrict__ tile_sums) {
    __shared__ u32 part[THREADS];
    const u32 l = blockIdx.y;
    const size_t base = (size_t)blockIdx.x * TILE + (size_t)threadIdx.x * ITEMS;
    u32 v[ITEMS], sum = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) { v[q] = base + q < n ? in[(size_t)l * n + base + q] : 0u; sum += v[q]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < THREADS; off <<= 1) {
        const u32 t = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    u32 run = part[threadIdx.x] - sum;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) { if (base + q < n) out[(size_t)l * n + base + q] = run; run += v[q]; }
    if (threadIdx.x == THREADS - 1) tile_sums[(size_t)l * gridDim.x + blockIdx.x] = part[THREADS - 1];
}
__global__ void __launch_bounds__(THREADS) scan_sums_kernel(u32* __restrict__ tile_sums, u32 ntiles, u32* __restrict__ totals) {
    __shared__ u32 part[THREADS];
    __shared__ u32 carry;
    const u32 l = blockIdx.x;
    u32* s = tile_sums + (size_t)l * ntiles;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 b0 = 0; b0 < ntiles; b0 += THREADS) {
        const u32 v = b0 + threadIdx.x < ntiles ? s[b0 + threadIdx.x] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < THREADS; off <<= 1) {
            const u32 t = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (b0 + threadIdx.x < ntiles) s[b0 + threadIdx.x] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == THREADS - 1) carry += part[THREADS - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[l] = carry;
}
// rows[l][pos] = i for the flagged positions, ascending (pos = tile-local exclusive scan + scanned tile sum)
__global__ void __launch_bounds__(256) compact_kernel(const u32* __restrict__ flags, const u32* __restrict__ pos, const u32* __restrict__ tile_sums, u32 ntiles, u32* __restrict__ rows, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 l = blockIdx.y;
    if (i < n && flags[(size_t)l * n + i]) rows[(size_t)l * n + pos[(size_t)l * n + i] + tile_sums[(size_t)l * ntiles + i / TILE]] = (u32)i;
}
// S'[row] = A'[row] at run starts; left-over table element k (ascending) goes to the (count - 1 - k)-th repeated row
__global__ void __launch_bounds__(256) place_table_kernel(const uint4* __restrict__ a_sorted, const u32* __restrict__ repeats, const uint4* __restrict__ table, size_t stride,
                                                          const u32* __restrict__ perm_tables, const u32* __restrict__ left_rows, const u32* __restrict__ rep_rows,
                                                          const u32* __restrict__ n_rep, const u32* __restrict__ n_left, uint4* __restrict__ out_table, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 l = blockIdx.y;
    if (i >= n) return;
    uint4* ot = out_table + 2 * (size_t)l * stride;
    if (!repeats[(size_t)l * n + i]) {
        const uint4* a = a_sorted + 2 * ((size_t)l * stride + i);
        ot[2 * i] = a[0]; ot[2 * i + 1] = a[1];
    }
    // repeated rows == left-over table elements (both usable_rows - #distinct inputs) unless an input value is missing from the table:
    // that lookup is reported as an error, its columns are not used -- only keep the indices inside the arrays
    const u32 count = n_rep[l] < n_left[l] ? n_rep[l] : n_left[l];
    if (i < count) {
        const uint4* src = table + 2 * ((size_t)l * stride + perm_tables[(size_t)l * n + left_rows[(size_t)l * n + i]]);
        const size_t dst = rep_rows[(size_t)l * n + count - 1 - i];
        ot[2 * dst] = src[0]; ot[2 * dst + 1] = src[1];
    }
}

struct Scratch {
    DevBuf planes, keys0, keys1, perm0, perm1, hist, flags, keep, pos, tile_sums, rows_rep, rows_left, small;
    u32* host = nullptr;  // pinned landing area of the flag words the host reads back per batch
    size_t host_words = 0;
};
Scratch& scratch() {  // per context (device)
    Ctx& c = ctx();
    if (!c.lookup_scratch) c.lookup_scratch = new Scratch();
    return *(Scratch*)c.lookup_scratch;
}

// the sort of `cols` columns: one step by the most significant varying limb (general = false), or one stable step per varying limb,
// least significant first (general = true).  Leaves the order in perm0.
int sort_columns(Scratch& sc, size_t n, u32 cols, bool general, u64* varies, int* key_limb, u64* key_mask, u32* side, u32* sel, hipStream_t s) {
    const unsigned gb = (unsigned)((n + 255) / 256), nblk = (unsigned)((n + TILE - 1) / TILE);
    u64 *keys0 = sc.keys0.as<u64>(), *keys1 = sc.keys1.as<u64>();
    u32 *perm0 = sc.perm0.as<u32>(), *perm1 = sc.perm1.as<u32>(), *hist = sc.hist.as<u32>();
    hipLaunchKernelGGL(reset_side_kernel, dim3((cols + 63) / 64), dim3(64), 0, s, side, cols);
    for (int step = general ? 0 : -1; step < (general ? 4 : 0); ++step) {
        hipLaunchKernelGGL(select_limb_kernel, dim3((cols + 63) / 64), dim3(64), 0, s, varies, cols, step, key_limb, key_mask, side);
        hipLaunchKernelGGL(gather_keys_kernel, dim3(gb, cols), dim3(256), 0, s, sc.planes.as<u64>(), n, cols, key_limb, side, perm0, perm1, keys0, keys1);
        for (int pass = 0; pass < PASSES; ++pass) {
            hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk, cols), dim3(THREADS), 0, s, keys0, keys1, side, key_mask, n, pass, hist);
            hipLaunchKernelGGL(radix_scan_kernel, dim3(cols), dim3(THREADS), 0, s, hist, nblk, key_mask, pass, side, sel);
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk, cols), dim3(THREADS), 0, s, keys0, keys1, perm0, perm1, keys0, keys1, perm0, perm1, sel, key_mask, n, pass, hist);
        }
    }
    hipLaunchKernelGGL(settle_perm_kernel, dim3(gb, cols), dim3(256), 0, s, perm0, perm1, side, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

// `batch` lookups at once; general: the all-limbs sort (the redo of a lookup whose fast sort met a tie between different values)
template <class F>
int lookup_permute_batch_t(const void* inputs, const void* tables, size_t n, size_t stride, u32 batch, void* out_inputs, void* out_tables, bool general, u32* bad_out /* batch, or null */,
                           hipStream_t s) {
    Scratch& sc = scratch();
    const u32 cols = 2 * batch;
    const size_t cn = (size_t)cols * n, bn = (size_t)batch * n;
    const unsigned gb = (unsigned)((n + 255) / 256), nblk = (unsigned)((n + TILE - 1) / TILE);
    TRH_TRY(sc.planes.ensure(cn * 32)); TRH_TRY(sc.keys0.ensure(cn * 8)); TRH_TRY(sc.keys1.ensure(cn * 8)); TRH_TRY(sc.perm0.ensure(cn * 4)); TRH_TRY(sc.perm1.ensure(cn * 4));
    TRH_TRY(sc.hist.ensure((size_t)cols * RADIX * nblk * 4)); TRH_TRY(sc.flags.ensure(bn * 4)); TRH_TRY(sc.keep.ensure(bn * 4)); TRH_TRY(sc.pos.ensure(bn * 4));
    TRH_TRY(sc.tile_sums.ensure((size_t)batch * nblk * 4)); TRH_TRY(sc.rows_rep.ensure(bn * 4)); TRH_TRY(sc.rows_left.ensure(bn * 4));
    // small per-column words: varies[cols][4] u64 | key_mask[cols] u64 | key_limb[cols] | side[cols] | sel[PASSES][cols] | bad[cols] | err[batch] | n_rep[batch] | n_left[batch]
    const size_t small_bytes = (size_t)cols * (32 + 8 + 4 + 4 + 4 * PASSES + 4) + (size_t)batch * 12 + 64;
    TRH_TRY(sc.small.ensure(small_bytes));
    char* sm = (char*)sc.small.p;
    u64* varies = (u64*)sm; sm += (size_t)cols * 32;
    u64* key_mask = (u64*)sm; sm += (size_t)cols * 8;
    int* key_limb = (int*)sm; sm += (size_t)cols * 4;
    u32* side = (u32*)sm; sm += (size_t)cols * 4;
    u32* sel = (u32*)sm; sm += (size_t)cols * 4 * PASSES;
    u32* bad = (u32*)sm; sm += (size_t)cols * 4;   // bad[cols], err[batch] contiguous: one read-back
    u32* err = (u32*)sm; sm += (size_t)batch * 4;
    u32* n_rep = (u32*)sm; sm += (size_t)batch * 4;
    u32* n_left = (u32*)sm;
    TRH_HIP_TRY(hipMemsetAsync(sc.small.p, 0, small_bytes, s));

    hipLaunchKernelGGL((canon_planes_kernel<F>), dim3(gb, cols), dim3(256), 0, s, (const uint4*)inputs, (const uint4*)tables, stride, batch, n, sc.planes.as<u64>(), sc.perm0.as<u32>());
    hipLaunchKernelGGL(plane_varies_kernel, dim3(gb, cols), dim3(256), 0, s, sc.planes.as<u64>(), n, varies);
    TRH_TRY(sort_columns(sc, n, cols, general, varies, key_limb, key_mask, side, sel, s));
    if (!general) hipLaunchKernelGGL(tie_check_kernel, dim3(gb, cols), dim3(256), 0, s, sc.planes.as<u64>(), n, sc.perm0.as<u32>(), key_limb, key_mask, sc.keys0.as<u64>(), sc.keys1.as<u64>(), side, bad);
    hipLaunchKernelGGL(gather_elems_kernel, dim3(gb, batch), dim3(256), 0, s, (const uint4*)inputs, stride, sc.perm0.as<u32>(), (uint4*)out_inputs, n);
    hipLaunchKernelGGL(repeat_flags_kernel, dim3(gb, batch), dim3(256), 0, s, (const uint4*)out_inputs, stride, n, sc.flags.as<u32>());
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((bn + 255) / 256)), dim3(256), 0, s, sc.keep.as<u32>(), bn, 1u);
    hipLaunchKernelGGL(remove_from_table_kernel, dim3(gb, batch), dim3(256), 0, s, sc.planes.as<u64>(), sc.perm0.as<u32>(), sc.flags.as<u32>(), n, batch, sc.keep.as<u32>(), err);
    // repeated input rows (ascending) and left-over table positions (ascending)
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(nblk, batch), dim3(THREADS), 0, s, sc.flags.as<u32>(), sc.pos.as<u32>(), n, sc.tile_sums.as<u32>());
    hipLaunchKernelGGL(scan_sums_kernel, dim3(batch), dim3(THREADS), 0, s, sc.tile_sums.as<u32>(), nblk, n_rep);
    hipLaunchKernelGGL(compact_kernel, dim3(gb, batch), dim3(256), 0, s, sc.flags.as<u32>(), sc.pos.as<u32>(), sc.tile_sums.as<u32>(), nblk, sc.rows_rep.as<u32>(), n);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(nblk, batch), dim3(THREADS), 0, s, sc.keep.as<u32>(), sc.pos.as<u32>(), n, sc.tile_sums.as<u32>());
    hipLaunchKernelGGL(scan_sums_kernel, dim3(batch), dim3(THREADS), 0, s, sc.tile_sums.as<u32>(), nblk, n_left);
    hipLaunchKernelGGL(compact_kernel, dim3(gb, batch), dim3(256), 0, s, sc.keep.as<u32>(), sc.pos.as<u32>(), sc.tile_sums.as<u32>(), nblk, sc.rows_left.as<u32>(), n);
    hipLaunchKernelGGL(place_table_kernel, dim3(gb, batch), dim3(256), 0, s, (const uint4*)out_inputs, sc.flags.as<u32>(), (const uint4*)tables, stride, sc.perm0.as<u32>() + bn, sc.rows_left.as<u32>(),
                       sc.rows_rep.as<u32>(), n_rep, n_left, (uint4*)out_tables, n);
    TRH_HIP_TRY(hipGetLastError());
    // one read-back per batch: tie flags of the 2 * batch columns, then the missing-value flags of the lookups
    const size_t words = (size_t)cols + batch;
    if (words > sc.host_words) {
        if (sc.host) (void)hipHostFree(sc.host);
        sc.host = nullptr;
        TRH_HIP_TRY(hipHostMalloc((void**)&sc.host, (words + 64) * 4, hipHostMallocDefault));
        sc.host_words = words + 64;
    }
    TRH_HIP_TRY(hipMemcpyAsync(sc.host, bad, words * 4, hipMemcpyDeviceToHost, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));
    for (u32 l = 0; l < batch; ++l) {
        const bool tie = sc.host[l] || sc.host[batch + l];
        if (bad_out) bad_out[l] = tie ? 1u : 0u;
        // a tie leaves the order of its members open, so a "missing" verdict of this pass is not final either: the redo decides
        if (!tie && sc.host[cols + l]) {
            set_error("lookup_permute: an input value of lookup %u does not occur in its table (halo2: Error::ConstraintSystemFailure)", l);
            return TRH_EINVAL;
        }
    }
    return TRH_OK;
}

template <class F>
int lookup_permute_all_t(const void* inputs, const void* tables, size_t n, size_t stride, size_t batch, void* out_inputs, void* out_tables, hipStream_t s) {
    // chunks of at most 32 lookups bound the scratch (2^18 rows: 32 lookups = 64 columns = 0.9 GiB)
    std::vector<u32> bad;
    for (size_t b0 = 0; b0 < batch; b0 += 32) {
        const u32 nb = (u32)(batch - b0 < 32 ? batch - b0 : 32);
        const char* in = (const char*)inputs + b0 * stride * 32;
        const char* tb = (const char*)tables + b0 * stride * 32;
        char* oi = (char*)out_inputs + b0 * stride * 32;
        char* ot = (char*)out_tables + b0 * stride * 32;
        bad.assign(nb, 0);
        TRH_TRY((lookup_permute_batch_t<F>(in, tb, n, stride, nb, oi, ot, false, bad.data(), s)));
        for (u32 l = 0; l < nb; ++l) {
            if (!bad[l]) continue;  // rare: two different values share the limb the fast path sorted by
            const int rc = lookup_permute_batch_t<F>(in + (size_t)l * stride * 32, tb + (size_t)l * stride * 32, n, stride, 1, oi + (size_t)l * stride * 32, ot + (size_t)l * stride * 32, true,
                                                     nullptr, s);
            if (rc != TRH_OK) {
                if (rc == TRH_EINVAL) set_error("lookup_permute: an input value of lookup %zu does not occur in its table (halo2: Error::ConstraintSystemFailure)", b0 + l);
                return rc;
            }
        }
    }
    return TRH_OK;
}

}  // namespace

void lookup_release() {
    Ctx& c = ctx();
    if (!c.lookup_scratch) return;
    Scratch& sc = *(Scratch*)c.lookup_scratch;
    for (DevBuf* b : {&sc.planes, &sc.keys0, &sc.keys1, &sc.perm0, &sc.perm1, &sc.hist, &sc.flags, &sc.keep, &sc.pos, &sc.tile_sums, &sc.rows_rep, &sc.rows_left, &sc.small}) b->release();
    if (sc.host) { (void)hipHostFree(sc.host); sc.host = nullptr; }
    delete &sc;
    c.lookup_scratch = nullptr;
}

}  // namespace trh

using namespace trh;

extern "C" {

int trh_lookup_permute_batch_dev(int field, const void* inputs_dev, const void* tables_dev, size_t usable_rows, size_t row_stride, size_t batch, void* out_inputs_dev,
                                 void* out_tables_dev, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (usable_rows && batch && (!inputs_dev || !tables_dev || !out_inputs_dev || !out_tables_dev)) { set_error("lookup_permute: null pointer"); return TRH_EINVAL; }
    if (out_inputs_dev == inputs_dev || out_tables_dev == tables_dev || out_inputs_dev == out_tables_dev) { set_error("lookup_permute: outputs must not alias the inputs"); return TRH_EINVAL; }
    if (usable_rows >= ((size_t)1 << 31) || row_stride < usable_rows) { set_error("lookup_permute: bad row count / stride"); return TRH_EINVAL; }
    if (!usable_rows || !batch) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_lookup_permute_batch_dev");
    if (field == TRH_FP) return lookup_permute_all_t<FpParams>(inputs_dev, tables_dev, usable_rows, row_stride, batch, out_inputs_dev, out_tables_dev, (hipStream_t)stream);
    return lookup_permute_all_t<FqParams>(inputs_dev, tables_dev, usable_rows, row_stride, batch, out_inputs_dev, out_tables_dev, (hipStream_t)stream);
}

int trh_lookup_permute_dev(int field, const void* input_dev, const void* table_dev, size_t usable_rows, void* out_input_dev, void* out_table_dev, void* stream) {
    return trh_lookup_permute_batch_dev(field, input_dev, table_dev, usable_rows, usable_rows, 1, out_input_dev, out_table_dev, stream);
}

}  // extern "C"
