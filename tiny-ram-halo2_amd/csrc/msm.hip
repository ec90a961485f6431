// Pippenger windowed-bucket multi-scalar multiplication over Pallas / Vesta for gfx950.
//
// Replaces halo2_proofs 0.2.0 `arithmetic::best_multiexp` / `multiexp_serial` (arithmetic.rs;
// crate pinned at /root/reference/Cargo.lock:619-621; reached from the reference through
// create_proof -> Params::commit / commit_lagrange and the IPA opening,
// /root/reference/src/test_utils.rs:41-49).  The result is the same group element; the
// schedule is built for the GPU instead of rayon chunks:
//
//   1. recode    one thread per scalar: Montgomery -> canonical (pasta `to_repr()`), signed
//                base-2^c digits (buckets 1..2^(c-1)), per-window bucket histogram
//   2. offsets   exclusive scan of the histogram per window
//   3. scatter   counting sort: point indices grouped by (window, bucket)
//   4. accumulate bases converted once to the signed lazy Montgomery domain (R'' = 2^261, 29-bit limbs, no modular
//                reduction in add/sub); one thread per fixed-length SEGMENT of the sorted list, so
//                every lane does the same number of XYZZ mixed adds (8M + 2S) whatever the bucket
//                sizes; per-bucket pieces are combined by one thread per bucket
//   5. reduce    sum_b b * B_b per window: per-thread running sums over a slice of buckets,
//                slice offset by a short double-and-add, LDS tree across the workgroup
//   6. combine   per-window sums -> host (W x 128 B), Horner over windows on the host
//
// Integer work, no MFMA.  Algorithmic HBM bytes: 32 B scalar + 64 B base per pair.
#include <stdlib.h>
#include <string.h>

#include "ctx.h"
#include "curve_q4.h"
#include "hostcombine.h"
#include "hosthelper.h"

namespace trh {

namespace {

constexpr int MAX_C = 18;
constexpr int MAX_C_FIXED = 18;  // fixed-base tables: one bucket set per MSM, so wider windows pay
constexpr u32 SIGN_BIT = 0x80000000u;

inline int ilog2_floor(size_t n) {
    int l = 0;
    while ((n >> (l + 1)) != 0) ++l;
    return l;
}

inline int choose_window_bits(size_t n) {
    int o = ctx().window_override;
    if (o >= 2 && o <= MAX_C) return o;
    // measured on MI355X (tools/window_sweep.py): below 2^15 pairs an MSM is latency-bound and the width
    // hardly matters; from 2^16 the wide windows win (fewer mixed adds, more level-1 sort bins)
    const int l = ilog2_floor(n ? n : 1);
    // 2^24 / 2^25: c = 17 (15 full windows + an almost always empty carry window) beats 16 by 3 %; from 2^26 the index
    // takes 26 of the 31 entry bits, which leaves 5 low bucket bits for the second sort level, and 16 wins again
    return l < 9 ? 4 : l < 15 ? 8 : l == 15 ? 11 : l == 16 ? 10 : l == 17 ? 12 : l < 20 ? 15 : l < 24 ? 16 : l < 26 ? 17 : 16;
}
inline int num_windows(int c) { return 255 / c + 1; }

// ---------------------------------------------------------------------------------------
// 1. recode: one thread per scalar (grid-stride).  Writes digits[j][i] = bucket | sign << 31
//    (bucket 0 = nothing to add) and a histogram over level-1 bins
//    bin = (bucket - 1) >> k2, accumulated in LDS and flushed once per workgroup.
// ---------------------------------------------------------------------------------------
template <class SF>
__global__ void __launch_bounds__(256) msm_recode_kernel(const uint4* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                         u32* __restrict__ digits, u32* __restrict__ bin_counts, int k2, u32 nbins, int use_lds, size_t sstride,
                                                         int one_row /* fixed-base mode: all windows share one histogram */,
                                                         const uint4* __restrict__ tails /* or null: scalar n - 1 of item z is tails[z] (the commitment blind) */,
                                                         unsigned char* __restrict__ tile_flags /* or null; fixed-base mode: [item][tile] = 1 when the partition tile holds an entry */,
                                                         u32 tile_log, u32 tiles) {
    const size_t z = blockIdx.z;  // batch item
    const u32 rows = one_row ? 1u : (u32)W;
    scalars += z * sstride * 2; digits += z * (size_t)W * n; bin_counts += z * (size_t)rows * nbins;
    extern __shared__ u32 lhist[];  // rows * nbins counters when use_lds
    const u32 total = rows * nbins;
    if (use_lds) {
        for (u32 k = threadIdx.x; k < total; k += blockDim.x) lhist[k] = 0;
        __syncthreads();
    }
    const u32 mask = (1u << c) - 1u, half = 1u << (c - 1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4* src = (tails && i == n - 1) ? tails + 2 * z : scalars + 2 * i;
        uint4 lo = src[0], hi = src[1];
        u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (mont) fe_store(fe_from_mont(fe_load<SF>(w)), w);
        u32 carry = 0;
        for (int j = 0; j < W; ++j) {
            u32 raw = (w[0] & mask) + carry;
            // shift the 256-bit value right by c (static register indexing)
#pragma unroll
            for (int k = 0; k < 7; ++k) w[k] = (w[k] >> c) | (w[k + 1] << (32 - c));
            w[7] >>= c;
            u32 bucket, sign = 0;
            if (raw > half) { bucket = (1u << c) - raw; carry = 1; sign = SIGN_BIT; }  // digit raw - 2^c
            else { bucket = raw; carry = 0; }
            digits[(size_t)j * n + i] = bucket ? (bucket | sign) : 0u;
            if (tile_flags) {  // witness columns leave most tiles of the flat W x n digit space empty: the partition skips those
                // one store per wave and window, no load: the lowest lane with an entry marks its tile; a lane within the first 64 slots of a
                // tile marks it too (the wave's lowest entry may lie in the tile before)
                const unsigned long long nz = __ballot(bucket != 0);
                const size_t slot = (size_t)j * n + i;
                const u32 lane = threadIdx.x & 63u;
                if (bucket && ((nz & ((1ull << lane) - 1ull)) == 0ull || (slot & (((size_t)1 << tile_log) - 1)) < 64))
                    tile_flags[z * tiles + (slot >> tile_log)] = 1;
            }
            if (bucket) {
                const u32 bin = (bucket - 1u) >> k2, row = one_row ? 0u : (u32)j;
                if (use_lds) atomicAdd(&lhist[row * nbins + bin], 1u);
                else atomicAdd(&bin_counts[(size_t)row * nbins + bin], 1u);
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (u32 k = threadIdx.x; k < total; k += blockDim.x) {
            const u32 v = lhist[k];
            if (v) atomicAdd(&bin_counts[k], v);
        }
    }
}

// up to four ranges cleared by ONE launch (a small MSM's pipeline cleared its counters with four hipMemsetAsync fill kernels of ~4.5 us each)
struct ZeroRanges { uint4* p[4]; u32 n16[4]; };
__global__ void __launch_bounds__(256) msm_zero_ranges_kernel(const ZeroRanges z) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < z.n16[k]; i += gridDim.x * blockDim.x) z.p[k][i] = zero;
}

// ---------------------------------------------------------------------------------------
// 2. exclusive scan per window of cnt[0..len) -> starts; cnt becomes the running cursor
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) msm_offsets_kernel(u32* __restrict__ counts, u32* __restrict__ starts, u32 nb1, u32* __restrict__ oversize, u32 bin_cap,
                                                           u32* __restrict__ totals /* or null: entries of this (item, row) */) {
    const size_t z = blockIdx.z;  // batch item
    counts += z * (size_t)gridDim.x * nb1; starts += z * (size_t)gridDim.x * nb1;
    __shared__ u32 part[1024];
    const int j = blockIdx.x, t = threadIdx.x;
    u32* cnt = counts + (size_t)j * nb1;
    u32* st = starts + (size_t)j * nb1;
    const u32 per = (nb1 + 1023u) / 1024u;
    const u32 lo = t * per < nb1 ? t * per : nb1, hi = (lo + per < nb1) ? lo + per : nb1;
    u32 sum = 0, biggest = 0;
    for (u32 b = lo; b < hi; ++b) { const u32 v = cnt[b]; sum += v; biggest = v > biggest ? v : biggest; }
    // a bin that does not fit the LDS of msm_bin_sort_kernel: this WINDOW takes the chunked passes instead (one flag per batch item and
    // window: the short top window of 254-bit scalars has bins twice the average, the other windows keep the LDS sort)
    if (oversize && biggest > bin_cap) atomicMax(oversize + z * gridDim.x + blockIdx.x, biggest);
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        u32 v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (totals && t == 1023) totals[z * gridDim.x + blockIdx.x] = part[1023];
    u32 run = part[t] - sum;
    for (u32 b = lo; b < hi; ++b) {
        u32 cv = cnt[b];
        st[b] = run;
        cnt[b] = run;  // cursor starts at the bin start
        run += cv;
    }
}

// ---------------------------------------------------------------------------------------
// 3a. partition: workgroup (tile, window) groups a tile of PART_TILE digits by level-1 bin.
//     LDS histogram of the tile -> one global atomicAdd per (workgroup, bin) reserves a run in the
//     bin's region -> entries are written into that run (rank from an LDS atomic), so each
//     workgroup writes ~PART_TILE / nbins consecutive entries per bin.
//     entry = point index | low k2 bucket bits << idx_bits | sign << 31
// ---------------------------------------------------------------------------------------
constexpr int PART_TILE = 16384;
constexpr int PART_THREADS = 1024;
__global__ void __launch_bounds__(PART_THREADS) msm_partition_kernel(const u32* __restrict__ digits, u32* __restrict__ bin_cursor,
                                                                     u32* __restrict__ parted, size_t n, int k2, u32 nbins, int idx_bits,
                                                                     const unsigned char* __restrict__ tile_flags /* or null: [item][tile], see msm_recode_kernel */) {
    const size_t z = blockIdx.z;  // batch item
    if (tile_flags && !tile_flags[z * gridDim.x + blockIdx.x]) return;  // no entry in this tile (uniform over the workgroup)
    digits += z * (size_t)gridDim.y * n; parted += z * (size_t)gridDim.y * n; bin_cursor += z * (size_t)gridDim.y * nbins;
    // LDS: the tile's entries staged in bin order (so that every bin's run leaves as one coalesced
    // store instead of PART_TILE scattered 4-byte writes), then per bin: count / cursor, start inside
    // the stage, start of the reserved run in HBM
    extern __shared__ u32 lds[];
    u32* stage = lds;
    u32* cnt = lds + PART_TILE;
    u32* lbase = cnt + nbins;
    u32* gbase = lbase + nbins;
    __shared__ u32 scan[PART_THREADS];
    const int j = blockIdx.y;
    const size_t t0 = (size_t)blockIdx.x * PART_TILE;
    const size_t t1 = t0 + PART_TILE < n ? t0 + PART_TILE : n;
    const u32* dg = digits + (size_t)j * n;
    for (u32 k = threadIdx.x; k < nbins; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    // the thread's PART_TILE / PART_THREADS digits stay in registers between the histogram and the scatter (one HBM read)
    constexpr int PER = PART_TILE / PART_THREADS;
    u32 mine[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const size_t i = t0 + threadIdx.x + (size_t)q * PART_THREADS;
        mine[q] = i < t1 ? dg[i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 b = mine[q] & ~SIGN_BIT;
        if (b) atomicAdd(&cnt[(b - 1u) >> k2], 1u);
    }
    __syncthreads();
    {   // exclusive scan of cnt over the bins -> lbase; one global atomic per bin reserves its run
        const u32 per = (nbins + PART_THREADS - 1) / PART_THREADS;
        const u32 lo = threadIdx.x * per < nbins ? threadIdx.x * per : nbins, hi = lo + per < nbins ? lo + per : nbins;
        u32 sum = 0;
        for (u32 k = lo; k < hi; ++k) sum += cnt[k];
        scan[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 1; off < PART_THREADS; off <<= 1) {
            const u32 v = ((int)threadIdx.x >= off) ? scan[threadIdx.x - off] : 0;
            __syncthreads();
            scan[threadIdx.x] += v;
            __syncthreads();
        }
        u32 run = scan[threadIdx.x] - sum;
        for (u32 k = lo; k < hi; ++k) {
            const u32 v = cnt[k];
            lbase[k] = run;
            gbase[k] = v ? atomicAdd(&bin_cursor[(size_t)j * nbins + k], v) : 0u;
            run += v;
        }
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < nbins; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    const u32 low_mask = (1u << k2) - 1u;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 e = mine[q];
        const u32 b = e & ~SIGN_BIT;
        if (b) {
            const u32 bm = b - 1u, bin = bm >> k2;
            const u32 r = atomicAdd(&cnt[bin], 1u);
            stage[lbase[bin] + r] = (u32)(t0 + threadIdx.x + (size_t)q * PART_THREADS) | ((bm & low_mask) << idx_bits) | (e & SIGN_BIT);
        }
    }
    __syncthreads();
    // copy out: one wave per bin, lanes cover consecutive entries of the run
    u32* out = parted + (size_t)j * n;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (u32 bin = wave; bin < nbins; bin += PART_THREADS / 64) {
        const u32 c = cnt[bin], lb = lbase[bin], gb = gbase[bin];
        for (u32 k = lane; k < c; k += 64) out[gb + k] = stage[lb + k];
    }
}

// ---------------------------------------------------------------------------------------
// 3b. bucket sort, chunk-parallel: the bin-grouped list is cut into fixed chunks of BS_CHUNK entries
//     whatever the bin sizes (witness-like scalars put millions of entries into one bin), and every
//     (chunk, bin) piece is ordered by the low k2 bucket bits:
//       count    LDS histogram per piece -> one global atomic per (piece, bucket)
//       ranges   per window exclusive scan of the bucket counts -> bucket ranges, write cursors and the
//                bucket of every segment start
//       scatter  per piece: LDS histogram -> one global atomic per (piece, bucket) reserves a run in the
//                bucket -> entries staged in LDS in bucket order -> coalesced copy-out (wave per bucket)
// ---------------------------------------------------------------------------------------
constexpr int BS_CHUNK = 8192;
constexpr int BS_THREADS = 512;

// first bin whose end lies beyond position pos (empty bins are skipped)
__device__ __forceinline__ u32 bin_of_position(const u32* __restrict__ bin_ends, u32 nbins, u32 pos) {
    u32 lo = 0, hi = nbins;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (bin_ends[mid] > pos) hi = mid; else lo = mid + 1;
    }
    return lo;
}

template <bool SCATTER>
__global__ void __launch_bounds__(BS_THREADS) msm_bucket_pass_kernel(const u32* __restrict__ parted, const u32* __restrict__ bin_starts,
                                                                     const u32* __restrict__ bin_ends, u32* __restrict__ bucket_cnt /* count: histogram; scatter: cursor */,
                                                                     u32* __restrict__ sorted, size_t n, int k2, u32 nbins, int idx_bits, u32 nbk,
                                                                     const u32* __restrict__ oversize) {
    const size_t z = blockIdx.z;  // batch item
    {
        const size_t Wz = gridDim.y;
        parted += z * Wz * n; sorted += z * Wz * n; bin_starts += z * Wz * nbins; bin_ends += z * Wz * nbins;
        bucket_cnt += z * Wz * (nbk + 1);
    }
    if (oversize && oversize[z * gridDim.y + blockIdx.y] == 0u) return;  // every bin of this window was sorted in LDS by msm_bin_sort_kernel
    __shared__ u32 cnt[128], tbase[128], gbase[128];
    __shared__ u32 stage[SCATTER ? BS_CHUNK : 1];
    const int j = blockIdx.y;
    const u32* bs = bin_starts + (size_t)j * nbins;
    const u32* be = bin_ends + (size_t)j * nbins;
    const u32 total = be[nbins - 1];  // bins are contiguous from 0: the last end is the entry count
    const u32 c0 = blockIdx.x * BS_CHUNK;
    if (c0 >= total) return;
    const u32 c1 = c0 + BS_CHUNK < total ? c0 + BS_CHUNK : total;
    const u32 nsub = 1u << k2, low_mask = nsub - 1u, idx_mask = (1u << idx_bits) - 1u;
    const u32* src = parted + (size_t)j * n;
    u32* dst = sorted + (size_t)j * n;
    u32* bc = bucket_cnt + (size_t)j * (nbk + 1);
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (u32 bin = bin_of_position(be, nbins, c0); bin < nbins; ++bin) {
        const u32 lo = bs[bin] > c0 ? bs[bin] : c0;
        if (lo >= c1) break;
        const u32 hi = be[bin] < c1 ? be[bin] : c1;
        if (hi <= lo) continue;  // empty bin
        if (hi - lo <= 64u) {
            // a handful of entries (the blinding rows of a witness column scatter ~100 entries over as many bins: the piece that holds them
            // paid four workgroup barriers per bin, 0.3 ms per batch): one global atomic per entry, no LDS, no barrier; uniform over the workgroup
            if (threadIdx.x < hi - lo) {
                const u32 e = src[lo + threadIdx.x];
                u32* cell = &bc[(bin << k2) + ((e >> idx_bits) & low_mask) + 1u];
                if (!SCATTER) atomicAdd(cell, 1u);
                else dst[atomicAdd(cell, 1u)] = (e & idx_mask) | (e & SIGN_BIT);
            }
            continue;
        }
        if (threadIdx.x < 128) cnt[threadIdx.x] = 0;
        __syncthreads();
        for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&cnt[(src[i] >> idx_bits) & low_mask], 1u);
        __syncthreads();
        if (!SCATTER) {
            if (threadIdx.x < nsub && cnt[threadIdx.x]) atomicAdd(&bc[(bin << k2) + threadIdx.x + 1u], cnt[threadIdx.x]);
        } else {
            if (threadIdx.x == 0) {
                u32 r = 0;
                for (u32 k = 0; k < nsub; ++k) { tbase[k] = r; r += cnt[k]; }
            }
            if (threadIdx.x < nsub) {
                const u32 v = cnt[threadIdx.x];
                gbase[threadIdx.x] = v ? atomicAdd(&bc[(bin << k2) + threadIdx.x + 1u], v) : 0u;  // reserve the run
            }
            __syncthreads();
            if (threadIdx.x < 128) cnt[threadIdx.x] = 0;
            __syncthreads();
            for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
                const u32 e = src[i];
                const u32 sub = (e >> idx_bits) & low_mask;
                const u32 r = atomicAdd(&cnt[sub], 1u);
                stage[tbase[sub] + r] = (e & idx_mask) | (e & SIGN_BIT);
            }
            __syncthreads();
            for (u32 sub = wave; sub < nsub; sub += BS_THREADS / 64) {
                const u32 c = cnt[sub], sb = tbase[sub], gb = gbase[sub];
                for (u32 k = lane; k < c; k += 64) dst[gb + k] = stage[sb + k];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// 3b'. bucket sort, whole bin in LDS: when every level-1 bin of the launch fits (random scalars: 2^15 +- 200 entries at 2^24
//      pairs), one workgroup sorts one bin -- entries into registers, LDS histogram over the <= 128 low bucket values, shuffle
//      scan, scatter into the LDS stage, ONE coalesced copy of the whole bin region -- and publishes the bucket ranges of its
//      bin itself (bin start + prefix), so the count pass, the global range scan and the per-piece run reservations of the
//      chunked passes are not needed.  An oversize bin (skewed scalars) flips `oversize` in msm_offsets_kernel and this kernel
//      returns; the chunked passes below, which return in the other case, then do the work.
// ---------------------------------------------------------------------------------------
constexpr int BIN_THREADS = 1024;
constexpr int BIN_PER_MAX = 36;                        // entries per thread, kept in registers from the load to the last pass
constexpr u32 BIN_CAP_MAX = BIN_THREADS * BIN_PER_MAX; // 36864 entries
// (Tried: a stage of half a bin filled in two passes, two 512-thread workgroups per CU so that one loads while the other
// scatters -- 72 entries per thread spill and the sort went from 1.61 to 2.29 ms at 2^24.  Round 2: a workgroup taking 8 consecutive
// bins with the NEXT bin's 36 entries per thread in flight in a second register array while the current bin is sorted in the LDS:
// 128 VGPRs + 132 B of scratch, sort 1.66 -> 1.73 ms -- the register arrays, not the unoverlapped loads, are what it costs.)
__global__ void __launch_bounds__(BIN_THREADS) msm_bin_sort_kernel(const u32* __restrict__ parted, const u32* __restrict__ bin_starts, const u32* __restrict__ bin_ends,
                                                                   u32* __restrict__ sorted, u32* __restrict__ starts, u32* __restrict__ ends, size_t n, int k2, u32 nbins,
                                                                   int idx_bits, u32 nbk, const u32* __restrict__ oversize) {
    if (oversize[(size_t)blockIdx.z * gridDim.y + blockIdx.y] != 0u) return;
    extern __shared__ u32 bstage[];
    __shared__ u32 cnt[128], tbase[128], wave0_total;
    const size_t z = blockIdx.z, Wz = gridDim.y;
    const int j = blockIdx.y;
    const u32 bin = blockIdx.x;
    const size_t wrow = z * Wz + j;
    const u32 lo = bin_starts[wrow * nbins + bin], hi = bin_ends[wrow * nbins + bin];
    const u32 count = hi - lo;
    const u32 nsub = 1u << k2, low_mask = nsub - 1u, idx_mask = (1u << idx_bits) - 1u;
    const u32* src = parted + wrow * n + lo;
    u32* dst = sorted + wrow * n + lo;
    const u32 lane = threadIdx.x & 63u;
    if (threadIdx.x < 128) cnt[threadIdx.x] = 0;
    __syncthreads();
    u32 mine[BIN_PER_MAX];
#pragma unroll
    for (int q = 0; q < BIN_PER_MAX; ++q) {
        const u32 i = threadIdx.x + (u32)q * BIN_THREADS;
        mine[q] = 0u;
        if (i < count) {
            mine[q] = src[i];
            atomicAdd(&cnt[(mine[q] >> idx_bits) & low_mask], 1u);
        }
    }
    __syncthreads();
    u32 v = 0, x = 0;
    if (threadIdx.x < 128) {  // exclusive scan of the sub-bucket counts inside the two waves that hold them
        v = threadIdx.x < nsub ? cnt[threadIdx.x] : 0u;
        x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off, 64);
            if ((int)lane >= off) x += y;
        }
        if (threadIdx.x == 63) wave0_total = x;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const u32 excl = x - v + (threadIdx.x >= 64 ? wave0_total : 0u);
        tbase[threadIdx.x] = excl;
        cnt[threadIdx.x] = 0;
        if (threadIdx.x < nsub) {  // bucket (bin << k2) + sub lives at index + 1 of the range arrays (index 0: digit 0, nothing to add)
            const size_t b = wrow * (nbk + 1) + ((size_t)bin << k2) + threadIdx.x + 1u;
            starts[b] = lo + excl;
            ends[b] = lo + excl + v;
        }
        if (bin == 0 && threadIdx.x == 0) { starts[wrow * (nbk + 1)] = 0; ends[wrow * (nbk + 1)] = 0; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BIN_PER_MAX; ++q) {
        if (threadIdx.x + (u32)q * BIN_THREADS < count) {
            const u32 e = mine[q];
            const u32 sub = (e >> idx_bits) & low_mask;
            const u32 r = atomicAdd(&cnt[sub], 1u);
            bstage[tbase[sub] + r] = (e & idx_mask) | (e & SIGN_BIT);
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < count; i += BIN_THREADS) dst[i] = bstage[i];
}

// per window: exclusive scan of the bucket counts -> starts / ends, cursor (in place of the counts).  Two launches over
// 1024-bucket blocks (coalesced): block totals, then every block adds the totals before it to its own LDS scan -- one
// workgroup per window walking 64 buckets per thread was a 0.18 ms latency chain at 2^16 buckets.
constexpr int RANGE_BLOCK = 1024;
__global__ void __launch_bounds__(RANGE_BLOCK) msm_bucket_block_sums_kernel(const u32* __restrict__ bucket_cnt, u32* __restrict__ block_sums, u32 nbk,
                                                                            const u32* __restrict__ oversize) {
    if (oversize && oversize[(size_t)blockIdx.z * gridDim.y + blockIdx.y] == 0u) return;
    __shared__ u32 part[RANGE_BLOCK / 64];
    const size_t z = blockIdx.z, Wz = gridDim.y;
    const u32 nb1 = nbk + 1, b = blockIdx.x * RANGE_BLOCK + threadIdx.x;
    u32 v = b < nb1 ? bucket_cnt[(z * Wz + blockIdx.y) * nb1 + b] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 0;
        for (int k = 0; k < RANGE_BLOCK / 64; ++k) t += part[k];
        block_sums[(z * Wz + blockIdx.y) * gridDim.x + blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(RANGE_BLOCK) msm_bucket_ranges_kernel(u32* __restrict__ bucket_cnt, const u32* __restrict__ block_sums, u32* __restrict__ starts,
                                                                        u32* __restrict__ ends, u32 nbk, const u32* __restrict__ oversize) {
    if (oversize && oversize[(size_t)blockIdx.z * gridDim.y + blockIdx.y] == 0u) return;
    __shared__ u32 part[RANGE_BLOCK];
    __shared__ u32 before;
    const size_t z = blockIdx.z, Wz = gridDim.y;
    const int t = threadIdx.x;
    const u32 nb1 = nbk + 1, b = blockIdx.x * RANGE_BLOCK + t;
    const size_t row = (z * Wz + blockIdx.y) * nb1;
    const u32 c = b < nb1 ? bucket_cnt[row + b] : 0u;
    {   // total of the blocks before this one (at most 2^18 / 1024 = 256 of them)
        u32 v = (u32)t < blockIdx.x ? block_sums[(z * Wz + blockIdx.y) * gridDim.x + t] : 0u;
        part[t] = v;
        __syncthreads();
        for (int s = RANGE_BLOCK / 2; s > 0; s >>= 1) {
            if (t < s) part[t] += part[t + s];
            __syncthreads();
        }
        if (t == 0) before = part[0];
        __syncthreads();
    }
    part[t] = c;
    __syncthreads();
    for (int off = 1; off < RANGE_BLOCK; off <<= 1) {  // Hillis-Steele inclusive scan
        const u32 v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (b < nb1) {
        const u32 E = before + part[t], S = E - c;
        starts[row + b] = S; ends[row + b] = E; bucket_cnt[row + b] = S;  // cursor starts at the bucket start
    }
}

// ---------------------------------------------------------------------------------------
// 4. accumulate
// ---------------------------------------------------------------------------------------
template <class BF>
__device__ __forceinline__ Affine<BF> load_affine(const uint4* __restrict__ bases, u32 idx) {
    const uint4* p = bases + (size_t)idx * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    Affine<BF> r;
    r.x = fe_load<BF>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
    r.y = fe_load<BF>(c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w);
    return r;
}
template <class BF>
__device__ __forceinline__ void store_fe4(uint4* p, const Fe<BF>& v) {
    u32 w[8];
    fe_store(v, w);
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
template <class BF>
__device__ __forceinline__ Fe<BF> load_fe4(const uint4* p) {
    uint4 a = p[0], b = p[1];
    return fe_load<BF>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
template <class BF>
__device__ __forceinline__ void store_xyzz(XYZZMem* dst, const XYZZ<BF>& v) {
    uint4* p = (uint4*)dst;
    store_fe4(p, v.x); store_fe4(p + 2, v.y); store_fe4(p + 4, v.zz); store_fe4(p + 6, v.zzz);
}
template <class BF>
__device__ __forceinline__ XYZZ<BF> load_xyzz(const XYZZMem* src) {
    const uint4* p = (const uint4*)src;
    XYZZ<BF> v;
    v.x = load_fe4<BF>(p); v.y = load_fe4<BF>(p + 2); v.zz = load_fe4<BF>(p + 4); v.zzz = load_fe4<BF>(p + 6);
    return v;
}

// ---------------------------------------------------------------------------------------
// 4. balanced accumulation in the lazy domain.
//   convert   bases (Montgomery R = 2^256) -> 128-byte records of ready limbs in the signed lazy domain (R'' = 2^261; ctx.h ZREC),
//             once per base set (owned handles keep them) or per MSM
//   segments  thread (segment, window) walks seg_len consecutive entries of the bucket-sorted
//             list: every lane does the same number of mixed adds whatever the bucket sizes.
//             At a bucket boundary the raw accumulator (36 limbs) is stored -- the first run of a
//             segment to first[], a run that ends past the segment to last[], a bucket that lies
//             inside the segment to direct[] -- and the accumulator restarts.
//   combine   thread per bucket: adds the pieces of its bucket (first[] of every segment it
//             covers, plus last[] / direct[] of the segment it starts in), converts to the
//             canonical Montgomery form and writes the bucket for the reduction kernels.
// ---------------------------------------------------------------------------------------
// x, y in the signed domain -> the 128-byte record (ctx.h ZREC)
template <class BF>
__device__ __forceinline__ void store_zrec(uint4* q, const Fy<BF>& x, const Fy<BF>& y) {
    q[0] = make_uint4((u32)x.l[0], (u32)x.l[1], (u32)x.l[2], (u32)x.l[3]); q[1] = make_uint4((u32)x.l[4], (u32)x.l[5], (u32)x.l[6], (u32)x.l[7]);
    q[2] = make_uint4((u32)y.l[0], (u32)y.l[1], (u32)y.l[2], (u32)y.l[3]); q[3] = make_uint4((u32)y.l[4], (u32)y.l[5], (u32)y.l[6], (u32)y.l[7]);
    // -y limb by limb: limbs in (-2^29, 0], as good as normalised ones wherever y is used (field.h "Signed lazy domain"); 0 stays 0
    q[4] = make_uint4((u32)-y.l[0], (u32)-y.l[1], (u32)-y.l[2], (u32)-y.l[3]); q[5] = make_uint4((u32)-y.l[4], (u32)-y.l[5], (u32)-y.l[6], (u32)-y.l[7]);
    q[6] = make_uint4((u32)x.l[8], (u32)y.l[8], (u32)-y.l[8], 0u);
    q[7] = make_uint4(0u, 0u, 0u, 0u);
}
template <class BF>
__global__ void __launch_bounds__(256) msm_convert_bases_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* p = in + i * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    // 0 -> 0: the identity stays (0, 0)
    store_zrec(out + i * (ZREC / 16), fy_from_fe(fe_load<BF>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w)), fy_from_fe(fe_load<BF>(c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w)));
}

template <class BF>
__device__ __forceinline__ void store_raw(XYZZzMem* dst, const XYZZz<BF>& v) {
    uint4* p = (uint4*)dst;
    const u32* w = (const u32*)&v;
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
template <class BF>
__device__ __forceinline__ XYZZz<BF> load_raw(const XYZZzMem* src) {
    const uint4* p = (const uint4*)src;
    XYZZz<BF> v;
    u32* w = (u32*)&v;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        uint4 q = p[k];
        w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
    }
    return v;
}

template <class BF>
__global__ void __launch_bounds__(256) msm_accumulate_seg_kernel(const uint4* __restrict__ bases_z, const u32* __restrict__ sorted,
                                                                 const u32* __restrict__ ends,
                                                                 XYZZzMem* __restrict__ first, XYZZzMem* __restrict__ last,
                                                                 XYZZzMem* __restrict__ direct, size_t n, u32 nbk, u32 nseg, u32 seg_len,
                                                                 const u32* __restrict__ lean_gate /* or null: lean sort -- a window whose bin sort gave up has no sorted list (msm_finish repeats the MSM) */) {
    const size_t z = blockIdx.z;  // batch item
    if (lean_gate && lean_gate[z * gridDim.y + blockIdx.y] != 0u) return;
    {
        const size_t Wz = gridDim.y;
        sorted += z * Wz * n; ends += z * Wz * (nbk + 1);
        first += z * Wz * nseg; last += z * Wz * nseg; direct += z * Wz * (nbk + 1);
    }
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (t >= nseg) return;
    const u32 nb1 = nbk + 1;
    const u32* en = ends + (size_t)j * nb1;
    const u32 total = en[nbk];  // entries of this window (zero digits are not listed)
    u32 pos = t * seg_len;
    if (pos >= total) return;
    const u32 stop = pos + seg_len < total ? pos + seg_len : total;
    // the bucket that holds the segment's first entry: smallest b with en[b] > pos (en is non-decreasing, en[0] = 0, pos < total = en[nbk]).
    // ~log2(nbk) L2-resident loads per thread (round 6: a kernel of its own until then -- 6 us of every small MSM's chain for the same loads)
    u32 B;
    {
        u32 lo = 0, hi = nbk;
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (en[mid] > pos) hi = mid; else lo = mid + 1;
        }
        B = lo;
    }
    u32 cur_end = en[B];
    const u32* lst = sorted + (size_t)j * n;
    XYZZzMem* my_first = first + (size_t)j * nseg + t;
    XYZZz<BF> acc = xyzzz_identity<BF>();
    bool is_first = true;
    bool fresh = true;  // acc holds no point of the current bucket yet (its limbs are the identity's, or stale)
    // two bases in flight: the gather for entry pos + 2 is issued while entry pos is added (its index was read one step earlier),
    // so neither the index read nor the gather of the record sits on the dependency chain of an iteration.  The look-ahead is clamped to
    // the segment's last entry instead of being conditional: a conditional load makes every slot register a merge of old and new
    // (nine 64-bit copies per step); the two repeated gathers per segment hit the L2.
    struct Slot { u32 e; uint4 a, b, c, d, t; };
    auto issue = [&](Slot& sl, u32 entry) {
        sl.e = entry;
        const uint4* bp = bases_z + (size_t)(entry & ~SIGN_BIT) * (ZREC / 16);
        const uint4* yp = bp + 2 + ((entry >> 31) << 1);  // y, or -y for a negative digit
        sl.a = bp[0]; sl.b = bp[1]; sl.c = yp[0]; sl.d = yp[1]; sl.t = bp[6];
    };
    auto unpack = [&](const Slot& sl, AffineZ<BF>& p) {
        p.x.l[0] = (i32)sl.a.x; p.x.l[1] = (i32)sl.a.y; p.x.l[2] = (i32)sl.a.z; p.x.l[3] = (i32)sl.a.w;
        p.x.l[4] = (i32)sl.b.x; p.x.l[5] = (i32)sl.b.y; p.x.l[6] = (i32)sl.b.z; p.x.l[7] = (i32)sl.b.w; p.x.l[8] = (i32)sl.t.x;
        p.y.l[0] = (i32)sl.c.x; p.y.l[1] = (i32)sl.c.y; p.y.l[2] = (i32)sl.c.z; p.y.l[3] = (i32)sl.c.w;
        p.y.l[4] = (i32)sl.d.x; p.y.l[5] = (i32)sl.d.y; p.y.l[6] = (i32)sl.d.z; p.y.l[7] = (i32)sl.d.w; p.y.l[8] = (i32)((sl.e >> 31) ? sl.t.z : sl.t.y);
    };
    const u32 last_pos = stop - 1;
    auto clamp = [&](u32 q) { return q < last_pos ? q : last_pos; };
    Slot s0, s1;
    issue(s0, lst[pos]);
    issue(s1, lst[clamp(pos + 1)]);
    u32 e_ahead = lst[clamp(pos + 2)];  // index of the entry two steps ahead
    auto step = [&](Slot& sl) {
        if (pos == cur_end) {  // bucket finished: publish its run and start the next bucket
            store_raw(is_first ? my_first : direct + (size_t)j * nb1 + B, acc);
            is_first = false;
            fresh = true;
            // next bucket that holds an entry: almost always the very next one; sparse lists (the few blinding rows of a witness column
            // scattered over 2^15 buckets) would otherwise walk thousands of empty buckets one dependent load at a time -- 1.3 ms of a
            // 4 ms batch of flag columns -- so after a few steps the search turns binary (en is non-decreasing)
            int walked = 0;
            do { ++B; cur_end = en[B]; } while (cur_end <= pos && ++walked < 4);
            if (cur_end <= pos) {
                u32 lo = B + 1, hi = nbk;  // smallest b with en[b] > pos; it exists because pos < total = en[nbk]
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if (en[mid] > pos) hi = mid; else lo = mid + 1;
                }
                B = lo;
                cur_end = en[B];
            }
        }
        const u32 ce = sl.e;
        AffineZ<BF> cur;
        unpack(sl, cur);
        // the identity is stored as (0, 0) and no point of these curves has y = 0 (odd prime order): y = 0 <=> identity
        const bool p_identity = ((cur.y.l[0] | cur.y.l[1] | cur.y.l[2]) | (cur.y.l[3] | cur.y.l[4] | cur.y.l[5]) | (cur.y.l[6] | cur.y.l[7] | cur.y.l[8])) == 0;
        issue(sl, e_ahead);
        e_ahead = lst[clamp(pos + 3)];
        if (!p_identity) {
            if (fresh) {  // first point of a bucket
                acc.x = cur.x; acc.y = cur.y; acc.zz = fy_one<BF>(); acc.zzz = fy_one<BF>();
                fresh = false;
            } else {
                Fy<BF> R;
                const bool same_x = xyzzz_madd_main(acc, cur, R);
                if (__any(same_x)) {  // wave-uniformly rare: acc was +-p.  The base is read again (keeping it live through the addition costs 18 registers)
                    if (same_x) {
                        if (fy_is_zero_mod(R)) {
                            Slot rs;
                            issue(rs, ce);
                            AffineZ<BF> again;
                            unpack(rs, again);
                            acc = xyzzz_dbl_affine(again);
                        } else {
                            acc = xyzzz_identity<BF>();
                            fresh = true;
                        }
                    }
                }
            }
        } else if (fresh) {
            acc = xyzzz_identity<BF>();  // stale limbs of the previous bucket must not be published if nothing else arrives
        }
        ++pos;
    };
    // pairs of steps without a condition between them (a conditional second step makes the loop's end state a merge of two
    // register sets: 130 copies per pair), then the odd one
    while (pos + 1 < stop) {
        step(s0);
        step(s1);
    }
    if (pos < stop) step(s0);
    if (is_first) store_raw(my_first, acc);
    else if (cur_end == stop) store_raw(direct + (size_t)j * nb1 + B, acc);
    else store_raw(last + (size_t)j * nseg + t, acc);
}
// (Round 6: holding this kernel to two waves per SIMD costs it nothing -- 14.24 against 14.20 ms at 2^24 -- but the sort and reduction kernels
// of neighbouring window groups that were to use the freed registers and LDS take their VALU issue slots and memory queue entries from it one
// for one: 17.7 against 16.9 ms per MSM.  profiles/r06_overlap_ab.txt; the pipeline is in the history at 780c803, not in the library.)
constexpr u32 HEAVY_PIECES = 64;

// buckets spanning more than HEAVY_PIECES segments (skewed scalars; the short top window of 255-bit
// scalars): one workgroup per listed bucket, threads stride over the pieces, LDS tree at the end
template <class BF>
__global__ void __launch_bounds__(256) msm_combine_heavy_kernel(const u32* __restrict__ starts, const u32* __restrict__ ends,
                                                                const XYZZzMem* __restrict__ first, const XYZZzMem* __restrict__ last,
                                                                XYZZzMem* __restrict__ buckets, u32 nbk, u32 nseg, u32 seg_len,
                                                                const u32* __restrict__ heavy, u32 W, u32 heavy_stride) {
    const size_t z = blockIdx.z;  // batch item
    {
        const size_t Wz = W;
        starts += z * Wz * (nbk + 1); ends += z * Wz * (nbk + 1); first += z * Wz * nseg; last += z * Wz * nseg;
        buckets += z * Wz * nbk; heavy += z * heavy_stride;
    }
    __shared__ XYZZz<BF> sh[256];
    const u32 count = heavy[0];
    const u32 nb1 = nbk + 1;
    for (u32 h = blockIdx.x; h < count; h += gridDim.x) {
        const u32 id = heavy[1 + h], j = id / nb1, b = id - j * nb1;
        const u32 S = starts[id], E = ends[id];
        const u32 t_lo = S / seg_len, t_hi = (E - 1) / seg_len;
        const XYZZzMem* fj = first + (size_t)j * nseg;
        XYZZz<BF> acc = xyzzz_identity<BF>();
        if (threadIdx.x == 0) acc = (S == t_lo * seg_len) ? load_raw<BF>(fj + t_lo) : load_raw<BF>(last + (size_t)j * nseg + t_lo);
        for (u32 t = t_lo + 1 + threadIdx.x; t <= t_hi; t += blockDim.x) acc = xyzzz_add(acc, load_raw<BF>(fj + t));
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) sh[threadIdx.x] = xyzzz_add(sh[threadIdx.x], sh[threadIdx.x + st]);
            __syncthreads();
        }
        if (threadIdx.x == 0) store_raw(&buckets[(size_t)j * nbk + (b - 1)], sh[0]);
        __syncthreads();
    }
}

// G lanes per bucket (1, 4 or 16): a small MSM with few, long buckets (the IPA rounds: 512 buckets of 512 entries in segments
// of 16) would otherwise add its 32 pieces one after the other in one thread -- a 0.15 ms latency chain per MSM; the lanes of
// a group take the pieces round-robin and a shuffle tree adds the group's partial sums
template <class BF, int G>
__global__ void __launch_bounds__(256) msm_combine_kernel(const u32* __restrict__ starts, const u32* __restrict__ ends,
                                                          const XYZZzMem* __restrict__ first, const XYZZzMem* __restrict__ last,
                                                          const XYZZzMem* __restrict__ direct, XYZZzMem* __restrict__ buckets,
                                                          u32 nbk, u32 nseg, u32 seg_len, u32* __restrict__ heavy /* [0] = count, then list */, u32 heavy_stride,
                                                          const u32* __restrict__ lean_gate /* or null, see msm_accumulate_seg_kernel */) {
    const size_t z = blockIdx.z;  // batch item
    if (lean_gate && lean_gate[z * gridDim.y + blockIdx.y] != 0u) return;
    {
        const size_t Wz = gridDim.y;
        starts += z * Wz * (nbk + 1); ends += z * Wz * (nbk + 1); first += z * Wz * nseg; last += z * Wz * nseg;
        direct += z * Wz * (nbk + 1); buckets += z * Wz * nbk; heavy += z * heavy_stride;
    }
    const u32 gt = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 b = gt / G + 1, sub = gt % G;  // nbk and the block size are multiples of G: a group is never split by the bound
    const int j = blockIdx.y;
    if (b > nbk) return;
    const u32 nb1 = nbk + 1;
    const u32 S = starts[(size_t)j * nb1 + b], E = ends[(size_t)j * nb1 + b];
    XYZZz<BF> acc = xyzzz_identity<BF>();
    if (E > S) {
        const u32 t_lo = S / seg_len, t_hi = (E - 1) / seg_len;
        if (t_hi - t_lo > HEAVY_PIECES) {  // skewed bucket: a whole workgroup adds its pieces (msm_combine_heavy_kernel)
            if (sub == 0) heavy[1 + atomicAdd(&heavy[0], 1u)] = (u32)j * nb1 + b;
            return;  // uniform over the group
        }
        const XYZZzMem* fj = first + (size_t)j * nseg;
        if (sub == 0) {
            if (S == t_lo * seg_len) acc = load_raw<BF>(fj + t_lo);
            else if (E <= (t_lo + 1) * seg_len) acc = load_raw<BF>(direct + (size_t)j * nb1 + b);
            else acc = load_raw<BF>(last + (size_t)j * nseg + t_lo);
        }
        for (u32 t = t_lo + 1 + sub; t <= t_hi; t += G) acc = xyzzz_add(acc, load_raw<BF>(fj + t));
    }
    if constexpr (G > 1) {
        for (int off = G / 2; off > 0; off >>= 1) {
            XYZZz<BF> o;
#pragma unroll
            for (int i = 0; i < NLIMBS; ++i) {
                o.x.l[i] = __shfl_down(acc.x.l[i], off, G); o.y.l[i] = __shfl_down(acc.y.l[i], off, G);
                o.zz.l[i] = __shfl_down(acc.zz.l[i], off, G); o.zzz.l[i] = __shfl_down(acc.zzz.l[i], off, G);
            }
            if ((int)sub + off < G) acc = xyzzz_add(acc, o);  // (a lane without a partner reads its own value back)
        }
        if (sub != 0) return;
    }
    store_raw(&buckets[(size_t)j * nbk + (b - 1)], acc);  // stays in the lazy domain for the reduction
}

// ---------------------------------------------------------------------------------------
// 5. reduce: window sum = sum_{b=1..nbk} b * B_b
//    thread t of a window owns buckets t*m+1 .. (t+1)*m; blocks of 256 threads tree-add in LDS.
// ---------------------------------------------------------------------------------------
template <class BF>
__global__ void __launch_bounds__(256) msm_reduce_kernel(const XYZZzMem* __restrict__ buckets, XYZZzMem* __restrict__ partials,
                                                         u32 nbk, u32 m, u32 threads_per_window) {
    const size_t z = blockIdx.z;  // batch item
    buckets += z * (size_t)gridDim.y * nbk; partials += z * (size_t)gridDim.y * gridDim.x;
    __shared__ XYZZz<BF> sh[256];  // register form (9 limbs per element), lazy domain throughout
    const int j = blockIdx.y;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZz<BF> total = xyzzz_identity<BF>();
    if (t < threads_per_window) {
        const XYZZzMem* bk = buckets + (size_t)j * nbk + (size_t)t * m;
        // Long slices (batches: 32 buckets per thread) of witness-shaped columns are almost empty -- a flag column fills about a hundred of
        // its 2^15 buckets --, and the running sums below walk them all: 2 m + ~15 dependent point operations.  A slice with at most four
        // live buckets takes sum_h id_h * B_h by ONE shared double-and-add over the bucket ids instead (15 doublings + ~7.5 additions per
        // live bucket, the points re-read from L1 where a bit is set): the kernel lasts as long as its slowest slice, and four is what the
        // fullest slice of such a batch holds
        u32 live = 0, kk[4] = {0, 0, 0, 0};
        if (m >= 16) {
            for (u32 k = 0; k < m; ++k) {
                const uint4* p = (const uint4*)&bk[k];
                const uint4 a = p[4], b = p[5], c2 = p[6];  // zz = words 18 .. 26: exactly zero <=> the bucket is the identity as the combine stored it
                if (a.z | a.w | b.x | b.y | b.z | b.w | c2.x | c2.y | c2.z) {
                    if (live < 4) kk[live] = k;
                    ++live;
                }
            }
        }
        if (m >= 16 && live <= 4) {
            XYZZz<BF> sc = xyzzz_identity<BF>();
            if (live) {
                const u32 base_id = t * m + 1;  // global id of the slice's first bucket
                for (int i = 31 - __clz(base_id + kk[live - 1]); i >= 0; --i) {  // the ids of a slice share their top bits: the largest one bounds the loop
                    sc = xyzzz_dbl(sc);
                    for (u32 h = 0; h < live; ++h)
                        if (((base_id + kk[h]) >> i) & 1u) sc = xyzzz_add(sc, load_raw<BF>(&bk[kk[h]]));
                }
            }
            total = sc;
        } else {
        XYZZz<BF> run = xyzzz_identity<BF>(), acc = xyzzz_identity<BF>();
        for (int k = (int)m - 1; k >= 0; --k) {
            const XYZZz<BF> v = load_raw<BF>(&bk[k]);
            run = xyzzz_add(run, v);
            acc = xyzzz_add(acc, run);
        }
        // acc = sum (k+1) * B[k]; the slice starts at global bucket id t*m + 1 -> add (t*m) * run
        const u32 off = t * m;
        if (off) {
            XYZZz<BF> sc = xyzzz_identity<BF>();
            int top = 31 - __clz(off);
            for (int i = top; i >= 0; --i) {
                sc = xyzzz_dbl(sc);
                if ((off >> i) & 1u) sc = xyzzz_add(sc, run);
            }
            acc = xyzzz_add(acc, sc);
        }
        total = acc;
        }
    }
    sh[threadIdx.x] = total;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = xyzzz_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) store_raw(&partials[(size_t)j * gridDim.x + blockIdx.x], sh[0]);
}

// The combine with a DPP quad per bucket (curve_q4.h): the quad adds the bucket's pieces one after the other at five multiplication steps each.
// The four-lanes-per-bucket form of msm_combine_kernel (G = 4) runs ceil(pieces / 4) + 2 FULL additions on every lane -- five for the three or
// four pieces a bucket of a lone commitment or an IPA round has --, the quad form pieces - 1 quad additions: the same lanes, a third of the work.
template <class BF>
__global__ void __launch_bounds__(256) msm_combine_q4_kernel(const u32* __restrict__ starts, const u32* __restrict__ ends,
                                                             const XYZZzMem* __restrict__ first, const XYZZzMem* __restrict__ last,
                                                             const XYZZzMem* __restrict__ direct, XYZZzMem* __restrict__ buckets,
                                                             u32 nbk, u32 nseg, u32 seg_len, u32* __restrict__ heavy /* [0] = count, then list */, u32 heavy_stride,
                                                             const u32* __restrict__ lean_gate /* or null, see msm_accumulate_seg_kernel */) {
    const size_t z = blockIdx.z;  // batch item
    if (lean_gate && lean_gate[z * gridDim.y + blockIdx.y] != 0u) return;
    {
        const size_t Wz = gridDim.y;
        starts += z * Wz * (nbk + 1); ends += z * Wz * (nbk + 1); first += z * Wz * nseg; last += z * Wz * nseg;
        direct += z * Wz * (nbk + 1); buckets += z * Wz * nbk; heavy += z * heavy_stride;
    }
    const u32 gt = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 b = gt / 4 + 1;
    const int q = threadIdx.x & 3;
    const int j = blockIdx.y;
    if (b > nbk) return;  // (nbk and the block size are multiples of four: a quad is never split by the bound)
    const u32 nb1 = nbk + 1;
    const u32 S = starts[(size_t)j * nb1 + b], E = ends[(size_t)j * nb1 + b];
    Fy<BF> acc = fy_zero<BF>();
    if (E > S) {
        const u32 t_lo = S / seg_len, t_hi = (E - 1) / seg_len;
        if (t_hi - t_lo > HEAVY_PIECES) {  // skewed bucket: a whole workgroup adds its pieces (msm_combine_heavy_kernel)
            if (q == 0) heavy[1 + atomicAdd(&heavy[0], 1u)] = (u32)j * nb1 + b;
            return;  // uniform over the quad
        }
        const XYZZzMem* fj = first + (size_t)j * nseg;
        if (S == t_lo * seg_len) acc = q4_load<BF>(fj + t_lo, q);
        else if (E <= (t_lo + 1) * seg_len) acc = q4_load<BF>(direct + (size_t)j * nb1 + b, q);
        else acc = q4_load<BF>(last + (size_t)j * nseg + t_lo, q);
        for (u32 t = t_lo + 1; t <= t_hi; ++t) acc = q4_add(acc, q4_load<BF>(fj + t, q), q);
    }
    q4_store(&buckets[(size_t)j * nbk + (b - 1)], q, acc);  // stays in the lazy domain for the reduction; an empty bucket is all zeros
}

// ---- the same reduction with every point operation spread over a DPP quad (curve_q4.h): for launches that are ONE latency chain ----------
// (a lone fixed-base commitment, an IPA round, a small MSM).  Quad t of a window owns the slice of m buckets the thread t of
// msm_reduce_kernel owns; the chain is the same (running sums, slice offset by double-and-add, workgroup tree over the 64 quads) at five /
// four multiplication steps per addition / doubling instead of fourteen / nine.  4 x the lanes for ~0.45 x the chain: only where lanes are idle.
template <class BF>
__global__ void __launch_bounds__(256) msm_reduce_q4_kernel(const XYZZzMem* __restrict__ buckets, XYZZzMem* __restrict__ partials, u32 nbk, u32 m, u32 quads_per_window) {
    const size_t z = blockIdx.z;  // batch item
    buckets += z * (size_t)gridDim.y * nbk; partials += z * (size_t)gridDim.y * gridDim.x;
    __shared__ Fy<BF> sh[256];  // coordinate q of quad Qi at [4 Qi + q]
    const int j = blockIdx.y;
    const int q = threadIdx.x & 3, Qi = threadIdx.x >> 2;
    const u32 t = blockIdx.x * 64u + (u32)Qi;
    Fy<BF> total = fy_zero<BF>();
    if (t < quads_per_window) {
        const XYZZzMem* bk = buckets + (size_t)j * nbk + (size_t)t * m;
        Fy<BF> run = fy_zero<BF>(), acc = fy_zero<BF>();
        for (int k = (int)m - 1; k >= 0; --k) {
            run = q4_add(run, q4_load<BF>(&bk[k], q), q);
            acc = q4_add(acc, run, q);
        }
        const u32 off = t * m;  // acc = sum (k + 1) B[k]; the slice starts at global bucket id t m + 1
        if (off) {
            Fy<BF> sc = fy_zero<BF>();
            for (int i = 31 - __clz(off); i >= 0; --i) {
                sc = q4_dbl(sc, q);
                if ((off >> i) & 1u) sc = q4_add(sc, run, q);
            }
            acc = q4_add(acc, sc, q);
        }
        total = acc;
    }
    sh[threadIdx.x] = total;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if (Qi < s) {  // reads [s, 2 s), writes [0, s): disjoint within a level
            total = q4_add(total, sh[((Qi + s) << 2) | q], q);
            sh[threadIdx.x] = total;
        }
        __syncthreads();
    }
    if (Qi == 0) q4_store(&partials[(size_t)j * gridDim.x + blockIdx.x], q, total);
}
// one block per window: the sum of `count` partials as 64 quads, handed over in the canonical form (lane q converts and stores coordinate q)
template <class BF>
__global__ void __launch_bounds__(256) msm_window_sum_q4_kernel(const XYZZzMem* __restrict__ partials, XYZZMem* __restrict__ window_sums, u32 count,
                                                                const u32* __restrict__ flag_src /* or null */, u32* __restrict__ flag_dst, u32 nflags) {
    // (the sort's "a bin did not fit the LDS" flags ride to the host behind the window sums: see `lean_sort` in msm_enqueue_t)
    if (flag_src && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x < nflags) flag_dst[threadIdx.x] = flag_src[threadIdx.x];
    const size_t z = blockIdx.z;
    partials += z * (size_t)gridDim.x * count; window_sums += z * (size_t)gridDim.x;
    __shared__ Fy<BF> sh[256];
    const int j = blockIdx.x;
    const int q = threadIdx.x & 3, Qi = threadIdx.x >> 2;
    Fy<BF> v = fy_zero<BF>();
    for (u32 k = (u32)Qi; k < count; k += 64) v = q4_add(v, q4_load<BF>(&partials[(size_t)j * count + k], q), q);
    sh[threadIdx.x] = v;
    __syncthreads();
    int top = 32;
    while (top > 1 && (u32)top >= count) top >>= 1;
    for (int s = top; s > 0; s >>= 1) {
        if (Qi < s) {
            v = q4_add(v, sh[((Qi + s) << 2) | q], q);
            sh[threadIdx.x] = v;
        }
        __syncthreads();
    }
    if (Qi == 0) {
        const bool id = q4_is_identity(v);
        const Fe<BF> c = id ? fe_zero<BF>() : fy_to_fe(v);
        store_fe4((uint4*)&window_sums[j] + 2 * q, c);
    }
}

// one block per window: sum `count` partials, hand the window sum over in the canonical form
template <class BF>
__global__ void __launch_bounds__(256) msm_window_sum_kernel(const XYZZzMem* __restrict__ partials, XYZZMem* __restrict__ window_sums, u32 count,
                                                             const u32* __restrict__ flag_src /* or null */, u32* __restrict__ flag_dst, u32 nflags) {
    if (flag_src && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x < nflags) flag_dst[threadIdx.x] = flag_src[threadIdx.x];
    const size_t z = blockIdx.z;  // batch item
    partials += z * (size_t)gridDim.x * count; window_sums += z * (size_t)gridDim.x;
    __shared__ XYZZz<BF> sh[256];
    const int j = blockIdx.x;
    XYZZz<BF> v = xyzzz_identity<BF>();
    for (u32 k = threadIdx.x; k < count; k += 256) v = xyzzz_add(v, load_raw<BF>(&partials[(size_t)j * count + k]));
    sh[threadIdx.x] = v;
    __syncthreads();
    int top = 128;  // only the levels that hold partials (16 of them for a lone 2^20 MSM: four levels instead of eight barriers + additions)
    while (top > 1 && (u32)top >= count) top >>= 1;
    for (int s = top; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = xyzzz_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) store_xyzz(&window_sums[j], xyzzz_to_canonical(sh[0]));
}

// ---------------------------------------------------------------------------------------
// synthetic bases: P_i = (s0 + (first + i) d) * G, G = (-1, 2)
// ---------------------------------------------------------------------------------------
template <class BF>
__global__ void __launch_bounds__(256) bases_generate_kernel(u64 s0, u64 d, u64 first, size_t n, uint4* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = s0 + (first + i) * d;
    Affine<BF> G;
    G.x = fe_neg(fe_one<BF>());
    G.y = fe_dbl(fe_one<BF>());
    XYZZ<BF> acc = xyzz_identity<BF>();
    for (int bit = 63; bit >= 0; --bit) {
        acc = xyzz_dbl(acc);
        if ((k >> bit) & 1ull) xyzz_madd(acc, G);
    }
    Affine<BF> a = xyzz_to_affine(acc);
    uint4* p = out + i * 4;
    store_fe4(p, a.x);
    store_fe4(p + 2, a.y);
}

// fixed-base table: out[j * n + i] = 2^(c j) * P_i in the lazy affine form (identity stays all-zero)
template <class BF>
__global__ void __launch_bounds__(256) msm_table_kernel(const uint4* __restrict__ bases, uint4* __restrict__ out, size_t n, int c, int W) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<BF> a;
    {
        const uint4* p = bases + i * 4;
        uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        a.x = fe_load<BF>(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w);
        a.y = fe_load<BF>(q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w);
    }
    for (int j = 0; j < W; ++j) {
        if (j) {
            XYZZ<BF> acc = xyzz_dbl_affine(a);
            for (int k = 1; k < c; ++k) acc = xyzz_dbl(acc);
            a = xyzz_to_affine(acc);
        }
        store_zrec(out + ((size_t)j * n + i) * (ZREC / 16), fy_from_fe(a.x), fy_from_fe(a.y));
    }
}

// ---------------------------------------------------------------------------------------
// Flag-like chunks of a fixed-base batch (round 4).  A witness column of the reference's circuit -- flags, <= 32-bit words, even-bits
// words, sorted small lookup values on n / 4 live rows, zero behind them, a handful of blinding rows -- has 3 * 10^4 .. 2 * 10^5
// non-zero digits in a flat digit space of W x n = 4 * 10^6 slots, and a FLAG column's digits are all +1: its commitment is a plain sum
// of table entries, which the bucket pipeline computes as one heavy bucket behind the whole sort / range / combine / reduce chain
// (1.6 - 2.5 ms per chunk of 64 such columns at k = 18, 0.1 ms of which is additions).
//   sample    one workgroup per column looks at ~1024 rows: a column whose estimated entry count is far above the list capacity is
//             dense; a column votes for the UNIT PATH when it is dense or shows nothing but 0 / +-1 digits.  The host reads the vote:
//             a chunk that is not unanimous runs the plain pipeline (sparse_chunk: the compact pipeline below measured no faster)
//   emit      one thread per scalar: canonical form, signed digits.  A digit +-1 never reaches a bucket -- the term is +-T[window][row]
//             itself -- and is appended to one of the column's SP_LISTS UNIT lists (flat table index | sign); every other non-zero
//             digit goes to one of its SP_LISTS compact digit lists as bucket | sign with its flat table index beside it (ONE atomic per
//             workgroup, window, kind and list: device-scope atomics are executed on the memory side of the eight XCDs' L2s and
//             same-address chains of them are slow); level-1 bin histogram in LDS as msm_recode_kernel has it.  A list that runs full
//             marks the column dense (the plain pipeline then recodes it from its scalars).
//   units     msm_unit_sum_kernel adds the listed table entries up (every thread a run of a list, an LDS tree per list)
//   tiny      what a flag column has beyond its ones are the ~100 digits of its blinding rows: a column with <= SP_TINY digit entries
//             skips the sort as well (digit * T[..] by a short double-and-add per entry); a chunk of such columns launches no pipeline
//   general   a column with more digits than the sampler saw: msm_offsets_kernel ... msm_bucket_pass_kernel over the compact digit
//             arrays (cap = W n / 8 slots per column), msm_sparse_remap_kernel replaces the sorted entries by the flat table indices
//             stored beside the digits, and from there the accumulate / combine / reduce launches of every other MSM
//   final     msm_unit_final_kernel: window sum = the partial sums of the lists (+ the tiny column's digit sum, + the pipeline's sum)
// Same group element, hence the same normalised point; which path a chunk or a column takes is decided by its digits alone.
// ---------------------------------------------------------------------------------------
constexpr int SP_LISTS = 16;              // compact lists per column (one counter each: a single counter per column would serialise its 1025 workgroups)
constexpr int SP_CNT = 2 * SP_LISTS;      // counters per column: the SP_LISTS digit lists, then the SP_LISTS unit lists
constexpr u32 SP_PAD = 32;                // u32 words between two list counters (one 128-byte line each)
constexpr u32 SP_TINY = 256;              // a column with at most this many digit entries is summed directly (no sort, no buckets)
constexpr int SP_PARTS = SP_LISTS + 1;    // partial sums per column: one per unit list, one for the digit entries of a tiny column
constexpr u32 SP_VOTES = 2;               // sp_count[SP_VOTES] (a spare word of the first counter line): columns of the chunk that vote for the unit path
constexpr u32 SP_DENSE = 0xFFFFFFFFu;     // counter 0 of a column the sampler found dense
constexpr size_t SP_MAX_CHUNK = 256;      // items per launch set the pinned read-back area is sized for

// scalar i of item z (the commitment blind for the last one) as canonical words
template <class SF>
__device__ __forceinline__ void sp_load_canonical(const uint4* __restrict__ scalars, const uint4* __restrict__ tails, size_t z, size_t i, size_t n, int mont, u32 w[8]) {
    const uint4* src = (tails && i == n - 1) ? tails + 2 * z : scalars + 2 * i;
    const uint4 lo = src[0], hi = src[1];
    w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w; w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
    if (mont && (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7])) fe_store(fe_from_mont(fe_load<SF>(w)), w);
}

template <class SF>
__global__ void __launch_bounds__(256) msm_sparse_sample_kernel(const uint4* __restrict__ scalars, size_t n, int mont, int c, int W, size_t sstride,
                                                                const uint4* __restrict__ tails, u32* __restrict__ sp_count, u32 dense_limit /* estimated entries above this: dense */) {
    const size_t z = blockIdx.x;
    scalars += z * sstride * 2;
    __shared__ u32 total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const size_t samples = n < 1024 ? n : 1024, step = n / samples;
    const u32 mask = (1u << c) - 1u, half = 1u << (c - 1);
    u32 mine = 0;  // non-zero digits, and (<< 16) those among them that are not +-1
    for (size_t q = threadIdx.x; q < samples; q += 256) {
        u32 w[8];
        sp_load_canonical<SF>(scalars, tails, z, q * step, n, mont, w);
        u32 carry = 0;
        for (int j = 0; j < W; ++j) {
            const u32 raw = (w[0] & mask) + carry;
#pragma unroll
            for (int k = 0; k < 7; ++k) w[k] = (w[k] >> c) | (w[k + 1] << (32 - c));
            w[7] >>= c;
            carry = raw > half ? 1u : 0u;
            const bool nonzero = raw != 0 && raw != (1u << c);  // digit raw - 2^c of raw == 2^c is zero with a carry
            mine += (nonzero ? 1u : 0u) + ((nonzero && raw != 1u && raw != (1u << c) - 1u) ? 0x10000u : 0u);
        }
    }
    atomicAdd(&total, mine);  // (at most 1024 x 16 digits: the halves cannot run into each other... 2^14 < 2^16)
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool dense = (unsigned long long)(total & 0xFFFFu) * step > dense_limit;
        if (dense) sp_count[z * SP_CNT * SP_PAD] = SP_DENSE;
        // the chunk's vote for the unit path (msm_sparse_emit_kernel): every column that is not dense shows nothing but 0 / +-1 digits
        if (dense || (total >> 16) == 0) atomicAdd(&sp_count[SP_VOTES], 1u);
    }
}

// digits: [item][cap] compact digit array (cap = SP_LISTS * subcap; list g of item z fills [g * subcap, ...)), zero-filled by the caller;
// flat: the flat table index (window * n + row) of every compact slot; bin_counts: [item][nbins] level-1 histogram (zeroed by the caller)
template <class SF>
__global__ void __launch_bounds__(256) msm_sparse_emit_kernel(const uint4* __restrict__ scalars, size_t n, int mont, int c, int W, size_t sstride,
                                                              const uint4* __restrict__ tails, u32* __restrict__ sp_count, u32* __restrict__ digits, u32* __restrict__ flat,
                                                              u32* __restrict__ units, u32 subcap, u32* __restrict__ bin_counts, int k2, u32 nbins) {
    const size_t z = blockIdx.z;
    u32* cnt0 = sp_count + z * SP_CNT * SP_PAD;
    if (*cnt0 == SP_DENSE) return;  // uniform over the grid slice of this column
    __shared__ u32 lhist[2048];     // nbins <= 2^11 (the partition's limit)
    __shared__ u32 wcnt[4], ucnt[4], base, ubase;
    const u32 g = blockIdx.x % SP_LISTS;
    u32* my = cnt0 + g * SP_PAD;
    u32* myu = cnt0 + (SP_LISTS + g) * SP_PAD;
    scalars += z * sstride * 2;
    digits += (z * SP_LISTS + g) * (size_t)subcap;
    flat += (z * SP_LISTS + g) * (size_t)subcap;
    units += (z * SP_LISTS + g) * (size_t)subcap;
    bin_counts += z * nbins;
    for (u32 k = threadIdx.x; k < nbins; k += blockDim.x) lhist[k] = 0;
    const u32 mask = (1u << c) - 1u, half = 1u << (c - 1);
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const bool unit_mode = sp_count[SP_VOTES] == gridDim.z;  // uniform over the grid: the sampler's votes are complete before this launch
    bool overflow = false;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x; i0 < n && !overflow; i0 += (size_t)gridDim.x * blockDim.x) {
        const size_t i = i0 + threadIdx.x;
        u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (i < n) sp_load_canonical<SF>(scalars, tails, z, i, n, mont, w);
        u32 carry = 0;
        for (int j = 0; j < W; ++j) {
            const u32 left = (w[0] | w[1] | w[2] | w[3]) | (w[4] | w[5] | w[6] | w[7]) | carry;
            if (!__syncthreads_or(left != 0)) break;  // small values: nothing above their top digit (uniform over the workgroup; also the barrier that frees wcnt / base)
            const u32 raw = (w[0] & mask) + carry;
#pragma unroll
            for (int k = 0; k < 7; ++k) w[k] = (w[k] >> c) | (w[k + 1] << (32 - c));
            w[7] >>= c;
            u32 bucket, sign = 0;
            if (raw > half) { bucket = (1u << c) - raw; carry = 1; sign = SIGN_BIT; }  // digit raw - 2^c
            else { bucket = raw; carry = 0; }
            const bool is_unit = unit_mode && bucket == 1u;          // +-1: the table entry itself
            const unsigned long long nz = __ballot(bucket != 0 && !is_unit);  // digits that need a bucket
            const unsigned long long nu = __ballot(is_unit);
            if (lane == 0) { wcnt[wv] = (u32)__popcll(nz); ucnt[wv] = (u32)__popcll(nu); }
            __syncthreads();
            const u32 c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3], total = c0 + c1 + c2 + c3;
            const u32 u0 = ucnt[0], u1 = ucnt[1], u2 = ucnt[2], u3 = ucnt[3], utotal = u0 + u1 + u2 + u3;
            if ((total | utotal) == 0) continue;  // uniform
            if (threadIdx.x == 0) {  // one atomic per workgroup, window and kind
                if (total) base = atomicAdd(my, total);
                if (utotal) ubase = atomicAdd(myu, utotal);
            }
            __syncthreads();
            const u32 b0 = total ? base : 0u, ub0 = utotal ? ubase : 0u;
            if (b0 + total > subcap || ub0 + utotal > subcap) { overflow = true; break; }  // uniform: a list is full, the column is dense (the host sees the counter)
            if (bucket != 0 && !is_unit) {
                const u32 pos = b0 + (wv > 0 ? c0 : 0u) + (wv > 1 ? c1 : 0u) + (wv > 2 ? c2 : 0u) + (u32)__popcll(nz & below);
                digits[pos] = bucket | sign;
                flat[pos] = (u32)((size_t)j * n + i);
                atomicAdd(&lhist[(bucket - 1u) >> k2], 1u);
            } else if (is_unit) {
                const u32 pos = ub0 + (wv > 0 ? u0 : 0u) + (wv > 1 ? u1 : 0u) + (wv > 2 ? u2 : 0u) + (u32)__popcll(nu & below);
                units[pos] = (u32)((size_t)j * n + i) | sign;
            }
        }
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < nbins; k += blockDim.x) {
        const u32 v = lhist[k];
        if (v) atomicAdd(&bin_counts[k], v);
    }
}

// what a column's counters say after the emit -- the same rule on the host (sparse_chunk) and in the kernels below:
//   dense    the sampler said so, or a list ran full: the plain pipeline recodes the column from its scalars, nothing emitted counts
//   tiny     at most SP_TINY digit entries: summed directly, never sorted
//   general  through the compact pipeline
enum { SP_CLASS_GENERAL = 0, SP_CLASS_DENSE = 1, SP_CLASS_TINY = 2 };
__host__ __device__ inline int sp_classify(const u32* cnt, u32 stride, u32 subcap) {  // cnt[k * stride], k < SP_CNT
    if (cnt[0] == SP_DENSE) return SP_CLASS_DENSE;
    size_t entries = 0;
    for (int k = 0; k < SP_CNT; ++k) {
        if (cnt[(size_t)k * stride] > subcap) return SP_CLASS_DENSE;
        if (k < SP_LISTS) entries += cnt[(size_t)k * stride];
    }
    return entries <= SP_TINY ? SP_CLASS_TINY : SP_CLASS_GENERAL;
}
// Tiny columns exist only on the unit path (`unit_mode`: the chunk's vote, known to the kernels and -- after the read-back -- to the host)
__host__ __device__ inline int sp_classify(const u32* cnt, u32 stride, u32 subcap, int unit_mode) {
    const int cls = sp_classify(cnt, stride, subcap);
    return (cls == SP_CLASS_TINY && !unit_mode) ? (int)SP_CLASS_GENERAL : cls;
}

// columns that do not go through the compact pipeline although they emitted something (a list overflowed: dense; on the unit path: few
// digits, summed directly): what they appended must not reach the sort
__global__ void __launch_bounds__(256) msm_sparse_neutralise_kernel(const u32* __restrict__ sp_count, u32 subcap, u32* __restrict__ digits, size_t cap, u32* __restrict__ bin_counts, u32 nbins) {
    const size_t z = blockIdx.y;
    const int unit_mode = sp_count[SP_VOTES] == gridDim.y;
    if (sp_classify(sp_count + z * SP_CNT * SP_PAD, SP_PAD, subcap, unit_mode) == SP_CLASS_GENERAL) return;  // uniform
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < cap; k += (size_t)gridDim.x * blockDim.x) digits[z * cap + k] = 0;
    if (blockIdx.x == 0) for (u32 k = threadIdx.x; k < nbins; k += blockDim.x) bin_counts[z * nbins + k] = 0;
}

// table record `flat` (x, and y or -y) as a lazy affine point; all-zero = identity
template <class BF>
__device__ __forceinline__ AffineZ<BF> sp_load_entry(const uint4* __restrict__ table, u32 entry) {
    const uint4* bp = table + (size_t)(entry & ~SIGN_BIT) * (ZREC / 16);
    const uint4* yp = bp + 2 + ((entry >> 31) << 1);
    const uint4 a = bp[0], b = bp[1], c = yp[0], d = yp[1], t = bp[6];
    AffineZ<BF> p;
    p.x.l[0] = (i32)a.x; p.x.l[1] = (i32)a.y; p.x.l[2] = (i32)a.z; p.x.l[3] = (i32)a.w;
    p.x.l[4] = (i32)b.x; p.x.l[5] = (i32)b.y; p.x.l[6] = (i32)b.z; p.x.l[7] = (i32)b.w; p.x.l[8] = (i32)t.x;
    p.y.l[0] = (i32)c.x; p.y.l[1] = (i32)c.y; p.y.l[2] = (i32)c.z; p.y.l[3] = (i32)c.w;
    p.y.l[4] = (i32)d.x; p.y.l[5] = (i32)d.y; p.y.l[6] = (i32)d.z; p.y.l[7] = (i32)d.w; p.y.l[8] = (i32)((entry >> 31) ? t.z : t.y);
    return p;
}

// Unit path only (the kernel returns at once otherwise).  part[z][g], g < SP_LISTS: the sum of the table entries named by unit list g of
// column z (every thread a contiguous run of the list, LDS tree); part[z][SP_LISTS]: the sum of digit * entry over ALL digit entries of a
// tiny column (one entry per thread and round: a double-and-add over the <= 15 bits of the digit), the identity for a general column.
// Dense columns: untouched (never read).
template <class BF>
__global__ void __launch_bounds__(256) msm_unit_sum_kernel(const uint4* __restrict__ table, const u32* __restrict__ sp_count, const u32* __restrict__ units,
                                                           const u32* __restrict__ digits, const u32* __restrict__ flat, u32 subcap, XYZZzMem* __restrict__ part) {
    if (sp_count[SP_VOTES] != gridDim.y) return;  // not on the unit path: nothing was diverted
    const size_t z = blockIdx.y;
    const u32 g = blockIdx.x;
    const u32* cnt = sp_count + z * SP_CNT * SP_PAD;
    __shared__ XYZZz<BF> sh[256];
    __shared__ int cls_s;
    if (threadIdx.x == 0) cls_s = sp_classify(cnt, SP_PAD, subcap, 1);
    __syncthreads();
    const int cls = cls_s;
    if (cls == SP_CLASS_DENSE) return;
    XYZZz<BF> acc = xyzzz_identity<BF>();
    if (g < (u32)SP_LISTS) {
        const u32 count = cnt[(SP_LISTS + g) * SP_PAD];
        const u32* lst = units + (z * SP_LISTS + g) * (size_t)subcap;
        const u32 per = (count + 255u) / 256u;
        const u32 lo = threadIdx.x * per, hi = lo + per < count ? lo + per : count;
        for (u32 k = lo; k < hi; ++k) xyzzz_madd(acc, sp_load_entry<BF>(table, lst[k]));
    } else if (cls == SP_CLASS_TINY) {
        for (int sub = 0; sub < SP_LISTS; ++sub) {
            const u32 count = cnt[sub * SP_PAD];
            const size_t off = (z * SP_LISTS + sub) * (size_t)subcap;
            for (u32 k = threadIdx.x; k < count; k += 256) {
                const u32 d = digits[off + k], bucket = d & ~SIGN_BIT;
                const AffineZ<BF> p = sp_load_entry<BF>(table, flat[off + k] | (d & SIGN_BIT));
                XYZZz<BF> r = xyzzz_identity<BF>();
                for (int i = 31 - __clz(bucket); i >= 0; --i) {  // bucket >= 2
                    r = xyzzz_dbl(r);
                    if ((bucket >> i) & 1u) xyzzz_madd(r, p);
                }
                acc = xyzzz_add(acc, r);
            }
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] = xyzzz_add(sh[threadIdx.x], sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) store_raw(&part[z * SP_PARTS + g], sh[0]);
}

// window sum of a sparse column.  Unit path: its SP_PARTS partial sums (+ what the compact pipeline left there for a general column).
// Otherwise the kernel only runs for a chunk whose sparse columns are all empty (no pipeline launched): the identity.
template <class BF>
__global__ void __launch_bounds__(64) msm_unit_final_kernel(const u32* __restrict__ sp_count, u32 subcap, const XYZZzMem* __restrict__ part, XYZZMem* __restrict__ window_sums,
                                                            int pipeline_ran) {
    const size_t z = blockIdx.x;
    const int unit_mode = sp_count[SP_VOTES] == gridDim.x;
    __shared__ XYZZz<BF> sh[32];
    __shared__ int cls_s;
    if (threadIdx.x == 0) cls_s = sp_classify(sp_count + z * SP_CNT * SP_PAD, SP_PAD, subcap, unit_mode);
    __syncthreads();
    const int cls = cls_s;
    if (cls == SP_CLASS_DENSE) return;  // the plain pipeline writes this column's sum
    const u32 t = threadIdx.x;
    if (t < 32) {
        XYZZz<BF> v = xyzzz_identity<BF>();
        if (t < (u32)SP_PARTS) { if (unit_mode) v = load_raw<BF>(&part[z * SP_PARTS + t]); }
        else if (t == (u32)SP_PARTS) { if (pipeline_ran && cls == SP_CLASS_GENERAL) v = xyzzz_from_canonical(load_xyzz<BF>(&window_sums[z])); }
        sh[t] = v;
    }
    __syncthreads();
    for (int st = 16; st > 0; st >>= 1) {
        if ((int)t < st) sh[t] = xyzzz_add(sh[t], sh[t + st]);
        __syncthreads();
    }
    if (t == 0) store_xyzz(&window_sums[z], xyzzz_to_canonical(sh[0]));
}

// sorted[z][k] names a slot of the compact digit array (| sign): replace it by that slot's flat table index
__global__ void __launch_bounds__(256) msm_sparse_remap_kernel(u32* __restrict__ sorted, const u32* __restrict__ flat, const u32* __restrict__ ends, size_t cap, u32 nbk) {
    const size_t z = blockIdx.y;
    const u32 total = ends[z * (nbk + 1) + nbk];
    sorted += z * cap; flat += z * cap;
    for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < total; k += gridDim.x * blockDim.x) {
        const u32 e = sorted[k];
        sorted[k] = flat[e & ~SIGN_BIT] | (e & SIGN_BIT);
    }
}


// ---------------------------------------------------------------------------------------
// Small MSMs in ONE launch (round 6).  Below ~2^13 pairs the pipeline above is a chain of ten launches whose kernels are each a few
// dependent point operations on a nearly empty chip (an IPA round over 2^12 + 2 points: 230 us of kernels).  Here a workgroup owns one
// (item, window): c = 5, 16 buckets x 16 lanes per bucket.
//   digits    the window's signed digit of every scalar, taken from v + H (H = sum (2^(c-1) - 1) 2^(c j): the unsigned digits of v + H are the
//             signed digits of v shifted by 2^(c-1) - 1 -- the same digit set as msm_recode_kernel's carry rule, without walking the windows
//             below), counting sort by bucket in LDS
//   add       the 16 lanes of a bucket split its list evenly and add their shares (mixed additions)
//   combine   shuffle tree over the 16 lanes; then sum_b b B_b as a suffix scan over the 16 buckets and a tree over the suffix sums:
//             8 + 4 full additions deep instead of 32 for running sums
// and writes the window sum where msm_finish expects it (host Horner over the 52 windows as for every per-window MSM).
// ---------------------------------------------------------------------------------------
constexpr int SMALL_C = 5;
constexpr u32 SMALL_NBK = 1u << (SMALL_C - 1);  // 16
constexpr u32 SMALL_MAX_N = 8448;               // LDS: one byte + one u16 per scalar
struct SmallBias { u32 w[9]; };

template <class BF>
__device__ __forceinline__ XYZZz<BF> shfl_down_point(const XYZZz<BF>& v, int off, int width) {
    XYZZz<BF> o;
#pragma unroll
    for (int l = 0; l < NLIMBS; ++l) {
        o.x.l[l] = __shfl_down(v.x.l[l], off, width); o.y.l[l] = __shfl_down(v.y.l[l], off, width);
        o.zz.l[l] = __shfl_down(v.zz.l[l], off, width); o.zzz.l[l] = __shfl_down(v.zzz.l[l], off, width);
    }
    return o;
}

template <class SF, class BF, bool Q4>
__global__ void __launch_bounds__(256) msm_small_kernel(const uint4* __restrict__ bases_z, const uint4* __restrict__ scalars, u32 n, int mont, size_t sstride,
                                                        const uint4* __restrict__ tails, const SmallBias H, XYZZMem* __restrict__ window_sums) {
    __shared__ signed char dig[SMALL_MAX_N];
    __shared__ unsigned short list[SMALL_MAX_N];
    __shared__ u32 hist[SMALL_NBK], cursor[SMALL_NBK], off[SMALL_NBK + 1];
    __shared__ XYZZzMem bsum[SMALL_NBK];
    const u32 j = blockIdx.x, z = blockIdx.y, W = gridDim.x, tid = threadIdx.x;
    scalars += (size_t)z * sstride * 2;
    if (tid < SMALL_NBK) hist[tid] = 0;
    __syncthreads();
    const u32 bit0 = SMALL_C * j, wi = bit0 >> 5, sh = bit0 & 31u;
    for (u32 i = tid; i < n; i += 256) {
        const uint4* src = (tails && i == n - 1) ? tails + 2 * z : scalars + 2 * (size_t)i;
        const uint4 lo = src[0], hi = src[1];
        u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (mont) fe_store(fe_from_mont(fe_load<SF>(w)), w);
        // v + H as nine words; the window's bits are in words wi, wi + 1
        u32 v[10];
        u64 cy = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { cy += (u64)w[k] + H.w[k]; v[k] = (u32)cy; cy >>= 32; }
        v[8] = (u32)cy + H.w[8]; v[9] = 0;
        u32 a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) if ((u32)k == wi) { a = v[k]; b = v[k + 1]; }
        const u32 raw = (u32)((((u64)b << 32) | a) >> sh) & ((1u << SMALL_C) - 1u);
        const int d = (int)raw - (int)(SMALL_NBK - 1);  // in [-15, 16]
        dig[i] = (signed char)d;
        if (d) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (u32 b = 0; b < SMALL_NBK; ++b) { off[b] = run; cursor[b] = run; run += hist[b]; }
        off[SMALL_NBK] = run;
    }
    __syncthreads();
    for (u32 i = tid; i < n; i += 256) {
        const int d = dig[i];
        if (d) list[atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u)] = (unsigned short)(i | (d < 0 ? 0x8000u : 0u));
    }
    __syncthreads();
    const u32 b = tid >> 4, sl = tid & 15u;
    XYZZz<BF> acc = xyzzz_identity<BF>();
    {
        const u32 start = off[b], cnt = off[b + 1] - start;
        const u32 lo = start + cnt * sl / 16, hi = start + cnt * (sl + 1) / 16;
        for (u32 e = lo; e < hi; ++e) {
            const u32 ent = list[e];
            const uint4* bp = bases_z + (size_t)(ent & 0x7FFFu) * (ZREC / 16);
            const u32 neg = ent >> 15;
            const uint4* yp = bp + 2 + (neg << 1);  // y, or -y for a negative digit
            const uint4 qa = bp[0], qb = bp[1], qc = yp[0], qd = yp[1], qt = bp[6];
            AffineZ<BF> p;
            p.x.l[0] = (i32)qa.x; p.x.l[1] = (i32)qa.y; p.x.l[2] = (i32)qa.z; p.x.l[3] = (i32)qa.w;
            p.x.l[4] = (i32)qb.x; p.x.l[5] = (i32)qb.y; p.x.l[6] = (i32)qb.z; p.x.l[7] = (i32)qb.w; p.x.l[8] = (i32)qt.x;
            p.y.l[0] = (i32)qc.x; p.y.l[1] = (i32)qc.y; p.y.l[2] = (i32)qc.z; p.y.l[3] = (i32)qc.w;
            p.y.l[4] = (i32)qd.x; p.y.l[5] = (i32)qd.y; p.y.l[6] = (i32)qd.z; p.y.l[7] = (i32)qd.w; p.y.l[8] = (i32)(neg ? qt.z : qt.y);
            xyzzz_madd(acc, p);  // (identity bases, P + P and P - P inside)
        }
    }
    if constexpr (!Q4) {
        for (int o = 8; o > 0; o >>= 1) {
            const XYZZz<BF> other = shfl_down_point(acc, o, 16);
            if ((int)sl < o) acc = xyzzz_add(acc, other);
        }
        if (sl == 0) store_raw(&bsum[b], acc);
        __syncthreads();
        if (tid >= 64) return;
        // the first wave: lane b < 16 holds bucket b + 1's sum; S_b = sum of the buckets >= b (suffix scan), total = sum of the S_b = sum (b + 1) B_b
        XYZZz<BF> v = tid < SMALL_NBK ? load_raw<BF>(&bsum[tid]) : xyzzz_identity<BF>();
        for (int o = 1; o < (int)SMALL_NBK; o <<= 1) {
            const XYZZz<BF> other = shfl_down_point(v, o, 16);
            if (tid < SMALL_NBK && (int)tid + o < (int)SMALL_NBK) v = xyzzz_add(v, other);
        }
        for (int o = SMALL_NBK / 2; o > 0; o >>= 1) {
            const XYZZz<BF> other = shfl_down_point(v, o, 16);
            if ((int)tid < o) v = xyzzz_add(v, other);
        }
        if (tid == 0) store_xyzz(&window_sums[(size_t)z * W + j], xyzzz_to_canonical(v));
    } else {
        // the same sums with a point per DPP quad (curve_q4.h: an addition is five multiplication steps deep instead of fourteen products one after
        // the other): 16 -> 8 partial sums per bucket in the plain form, those through LDS into quads -- 64 pairs on 256 lanes --, two more levels by
        // shuffles between the quads of a bucket; then the 16 bucket sums on the 16 quads of the first wave: suffix scan + tree as above
        __shared__ XYZZzMem pts[SMALL_NBK * 8];
        {
            const XYZZz<BF> other = shfl_down_point(acc, 8, 16);
            if (sl < 8) { acc = xyzzz_add(acc, other); store_raw(&pts[b * 8 + sl], acc); }
        }
        __syncthreads();
        const int q = (int)(tid & 3u);
        const u32 g = tid >> 2;  // quad: bucket g >> 2, pair g & 3
        auto shfl_fy = [](const Fy<BF>& a, int lanes) {
            Fy<BF> r;
#pragma unroll
            for (int l = 0; l < NLIMBS; ++l) r.l[l] = __shfl_down(a.l[l], lanes);
            return r;
        };
        Fy<BF> v = q4_add(q4_load<BF>(&pts[(g >> 2) * 8 + (g & 3u)], q), q4_load<BF>(&pts[(g >> 2) * 8 + (g & 3u) + 4], q), q);
        // (a quad without a partner adds the identity: a lane past the wave's end would read itself back, and P + P takes the doubling's extra steps)
        v = q4_add(v, q4_select((g & 3u) < 2, shfl_fy(v, 8), fy_zero<BF>()), q);   // pairs 0 and 1 of a bucket
        v = q4_add(v, q4_select((g & 3u) == 0, shfl_fy(v, 4), fy_zero<BF>()), q);  // pair 0: the bucket's sum
        if ((g & 3u) == 0) q4_store(&bsum[g >> 2], q, v);
        __syncthreads();
        if (tid >= 64) return;
        v = q4_load<BF>(&bsum[g], q);  // quad g < 16 of the first wave: bucket g + 1
        for (int o = 1; o < (int)SMALL_NBK; o <<= 1) v = q4_add(v, q4_select((int)g + o < (int)SMALL_NBK, shfl_fy(v, 4 * o), fy_zero<BF>()), q);
        for (int o = SMALL_NBK / 2; o > 0; o >>= 1) v = q4_add(v, q4_select((int)g < o, shfl_fy(v, 4 * o), fy_zero<BF>()), q);
        if (g == 0) {  // lane q holds coordinate q of the window sum
            const bool id = q4_is_identity(v);
            uint4* dst = (uint4*)&window_sums[(size_t)z * W + j] + 2 * q;
            if (id) { dst[0] = make_uint4(0, 0, 0, 0); dst[1] = make_uint4(0, 0, 0, 0); }
            else store_fe4(dst, fy_to_fe(v));
        }
    }
}

template <class BF> int msm_finish_t(hipStream_t s, u64* out_xyz, size_t batch);
template <class BF> int point_sum_host_t(const u64* pts, size_t count, u64* out);

// Above 2^25 pairs the entry index squeezes the second sort level (31 bits = index + low bucket bits) and the rate drops
// (2^25: 815 M pairs/s, 2^26: 740, 2^27: 490, 2^28: 410): larger MSMs run as equal range tiles of at most 2^25 pairs, each
// at the full rate; the tiles' points are added on the host (the same sum the range-sharded multi-GPU path forms).
constexpr size_t MSM_TILE = (size_t)1 << 25;

template <class SF, class BF>
int msm_enqueue_t(const void* bases_dev, const void* bases_z, const void* scalars_dev, size_t n, size_t batch, size_t stride, int mont, hipStream_t s, const MsmFixedBase* fb,
                  const void* tails_dev) {
    Ctx& c = ctx();
    MsmScratch& m = c.msm;
    MsmLane& L = m.lane;
    if (fb && (n == 0 || !msm_fixed_base_fits(n, fb->c))) fb = nullptr;
    if (!m.in_tile) m.tile_sum_valid = false;
    if (!fb && batch == 1 && n > MSM_TILE && !m.in_tile && c.window_override == 0) {
        const size_t tiles = (n + MSM_TILE - 1) / MSM_TILE, len = (n + tiles - 1) / tiles;
        u64 acc[24];
        memset(acc, 0, sizeof(acc));
        m.in_tile = true;
        int rc = TRH_OK;
        for (size_t off = 0; off < n && rc == TRH_OK; off += len) {
            const size_t cur = off + len < n ? len : n - off;
            const bool last = off + cur == n;
            rc = msm_enqueue_t<SF, BF>((const char*)bases_dev + off * 64, bases_z ? (const char*)bases_z + off * ZREC : nullptr, (const char*)scalars_dev + off * 32, cur, 1, cur,
                                       mont, s, nullptr, last ? tails_dev : nullptr);
            if (rc != TRH_OK || last) break;
            rc = msm_finish_t<BF>(s, acc + 12, 1);  // blocks: the next tile reuses the sort scratch anyway
            if (rc == TRH_OK) rc = point_sum_host_t<BF>(acc, 2, acc);
        }
        m.in_tile = false;
        if (rc != TRH_OK) return rc;
        memcpy(m.tile_sum, acc, 96);
        m.tile_sum_valid = true;
        return TRH_OK;
    }
    if (!fb && n != 0 && n <= SMALL_MAX_N && batch <= 4 && c.window_override == 0 && !m.in_tile && !m.force_fallback) {  // one launch (msm_small_kernel)
        const int W = num_windows(SMALL_C);
        const size_t hs = batch * W * sizeof(XYZZMem);
        TRH_TRY(m.window_sums.ensure(hs));
        if (hs > m.host_sums_cap) {
            if (m.host_sums) (void)hipHostFree(m.host_sums);
            TRH_HIP_TRY(hipHostMalloc(&m.host_sums, hs + 4096, hipHostMallocDefault));
            m.host_sums_cap = hs + 4096;
        }
        if (!bases_z) TRH_TRY(m.bases_z.ensure(n * ZREC + ZREC));
        if (m.reserve_only) return TRH_OK;
        if (!bases_z) {
            hipLaunchKernelGGL((msm_convert_bases_kernel<BF>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint4*)bases_dev, m.bases_z.as<uint4>(), n);
            bases_z = m.bases_z.p;
        }
        SmallBias H{};  // sum over the windows of (2^(c-1) - 1) 2^(c j), 9 x 32 bits
        for (int j = 0; j < W; ++j) {
            const u64 v = (u64)(SMALL_NBK - 1) << ((SMALL_C * j) & 31);
            const int k = (SMALL_C * j) >> 5;
            if (k < 9) H.w[k] |= (u32)v;
            if (k + 1 < 9) H.w[k + 1] |= (u32)(v >> 32);
        }
        if (opt().reduce_q4)
            hipLaunchKernelGGL((msm_small_kernel<SF, BF, true>), dim3((unsigned)W, (unsigned)batch), dim3(256), 0, s, (const uint4*)bases_z, (const uint4*)scalars_dev, (u32)n, mont, stride,
                               (const uint4*)tails_dev, H, m.window_sums.as<XYZZMem>());
        else
            hipLaunchKernelGGL((msm_small_kernel<SF, BF, false>), dim3((unsigned)W, (unsigned)batch), dim3(256), 0, s, (const uint4*)bases_z, (const uint4*)scalars_dev, (u32)n, mont, stride,
                               (const uint4*)tails_dev, H, m.window_sums.as<XYZZMem>());
        TRH_HIP_TRY(hipGetLastError());
        TRH_HIP_TRY(hipMemcpyAsync(m.host_sums, m.window_sums.p, hs, hipMemcpyDeviceToHost, s));
        m.pending_curve = BF::ID; m.pending_windows = W; m.pending_c = SMALL_C; m.pending_batch = batch; m.pending_stream = s; m.pending_owner = nullptr;
        m.ev_valid = false; m.lean_pending = false;
        ++m.small_launches;
        return TRH_OK;
    }
    int cb = fb ? fb->c : choose_window_bits(n);
    if (!fb) {
        // beyond 2^27 pairs the index leaves fewer than 4 entry bits for the second sort level; the first level has at most
        // 2^11 bins (LDS of the partition), so the window narrows with n (15 bits up to 2^28 pairs ... 12 up to 2^31)
        int ib = 1;
        while (((size_t)1 << ib) < n) ++ib;
        const int k2max = 31 - ib < 7 ? 31 - ib : 7;
        if (cb - 1 - k2max > 11) cb = 12 + k2max;
    }
    const int W = fb ? fb->W : num_windows(cb);  // windows of the recoding
    // fixed-base mode: the W x n digits are one flat list over the W x n table entries -> ONE bucket set
    const int Ws = fb ? 1 : W;
    const size_t ns = fb ? (size_t)W * n : n;
    const u32 nbk = 1u << (cb - 1), nb1 = nbk + 1;
    // sort geometry: bucket - 1 = bin << k2 | sub; the partitioned entry packs sub above the index
    int idx_bits = 1;
    while (((size_t)1 << idx_bits) < ns) ++idx_bits;
    int k2 = cb - 1 < 7 ? cb - 1 : 7;
    if (k2 > 31 - idx_bits) k2 = 31 - idx_bits;
    const int k1 = cb - 1 - k2;
    const u32 nbins = 1u << k1;
    const size_t recode_lds = (size_t)Ws * nbins * 4;
    const int recode_use_lds = recode_lds <= 64 * 1024;
    // independent batch items (one MSM per column of create_proof, same bases) are processed
    // `chunk` at a time by the SAME launches (blockIdx.z = item), so the latency-bound sort and
    // reduction phases of one item are hidden behind the work of the others
    size_t chunk = batch;
    {
        const size_t per_item = (size_t)W * n * 12 + 1;  // digits + parted + sorted dominate
        // at most 64 items per launch set, inside the scratch budget (option msm_chunk_gb, default 4 GiB per digit array set)
        const size_t cap = ((size_t)opt().msm_chunk_gb << 30) / per_item;
        if (chunk > cap) chunk = cap ? cap : 1;
        if (chunk > 64) chunk = 64;
        static_assert(64 <= SP_MAX_CHUNK, "chunk size against the pinned read-back area");
    }
    // reduce geometry: each thread owns a slice of buckets and pays one short scalar multiplication for
    // the slice offset, so long slices do less work per bucket but are a long serial chain: a lone MSM
    // (latency-bound) gets 2048-4096 threads per window, a batch (throughput-bound) as few as 256
    u32 tpw = nbk >= (1u << 15) ? 4096 : 2048;  // slices of >= 8 buckets (measured: 2^20..2^24 pairs gain 0.07-0.16 ms, 2^18 loses with 4096)
    if (fb) tpw = 16384;  // one flat window per item: slices of 2 buckets while the batch is small (the cap below takes over for batches) -- the
                          // opening's rounds are two such items each: reduce 240 -> 190 us per round, k = 18 opening 15.4 -> 14.6 ms
    while (tpw > 256 && (size_t)Ws * tpw * chunk > ((size_t)1 << 16)) tpw >>= 1;  // 2^16 threads = one wave per SIMD (batch of 64 commits: 0.79 -> 0.62 ms)
    if (tpw > nbk) tpw = nbk;
    const u32 slice = nbk / tpw;
    const u32 rblocks = (tpw + 255) / 256;
    // segment length: enough segments to fill the chip (>= ~2^18 threads) but at most 128 entries each
    u32 seg_len0 = 128;
    // (2^18, round 6: an opening's full-size round -- 2 x 4.2 M digit slots, half of them empty -- takes segments of 32 instead of 16: four pieces per
    //  bucket for the combine instead of eight, k = 18 opening 8.2 -> 8.0 ms; 2^20 .. 2^22 MSMs and lone commitments unchanged; 2^17 loses: 9.0 ms)
    while (seg_len0 > 16 && (size_t)W * n * chunk / seg_len0 < ((size_t)1 << 18)) seg_len0 >>= 1;  // W * n == Ws * ns
    const u32 nseg0 = (u32)((ns + seg_len0 - 1) / seg_len0);
    // a heavy bucket spans > HEAVY_PIECES segments, so there are fewer than W * nseg / HEAVY_PIECES of them
    const size_t max_heavy = (size_t)Ws * nseg0 / HEAVY_PIECES + 1;
    const u32 heavy_stride0 = (u32)(max_heavy + 1);
    const unsigned heavy_blocks0 = (unsigned)(max_heavy < 256 ? max_heavy : 256);

    TRH_TRY(L.digits.ensure(chunk * W * n * 4 + 16));
    TRH_TRY(L.parted.ensure(chunk * W * n * 4 + 16));
    TRH_TRY(L.sorted.ensure(chunk * W * n * 4 + 16));
    // L.counts: [bin counts][fixed-base mode: one byte per (item, partition tile)][oversize-bin flags of the LDS bin sort][entry totals per item and window]
    const u32 part_tiles = (u32)((ns + PART_TILE - 1) / PART_TILE);
    const size_t flag_bytes = fb ? ((size_t)chunk * part_tiles + 3) / 4 * 4 : 0;
    TRH_TRY(L.counts.ensure(chunk * Ws * nbins * 4 + flag_bytes + 2 * chunk * Ws * 4 + 16));
    unsigned char* const tile_flags = fb ? (unsigned char*)(L.counts.as<u32>() + chunk * Ws * nbins) : nullptr;
    u32* const oversize = (u32*)((char*)L.counts.p + chunk * Ws * nbins * 4 + flag_bytes);
    u32* const totals = oversize + chunk * Ws;
    // Batched commitments of WITNESS columns (flags, small words: a few 10^4 entries per column, most of them in a handful of buckets)
    // leave the sorted lists almost empty, and fixed 128-entry segments then mean a few hundred threads each walking a serial chain of 128
    // mixed additions (3 ms per batch of 64 flag columns at k = 18, the chip idle).  For batches the entry counts are read back after the
    // histogram scan (one synchronisation per chunk, ~20 us) and the segment length is sized to the entries that exist.
    const bool adaptive = batch >= 8;
    if (adaptive && !c.pinned_land) TRH_HIP_TRY(hipHostMalloc(&c.pinned_land, 4096, hipHostMallocDefault));  // (>= SP_MAX_CHUNK x 16 windows x 4 B)
    // LDS bin sort when the bins are big enough to fill a 1024-thread workgroup and fit with 6 % + 512 entries of slack
    // (uniform digits: the largest of 8192 bins of 2^15 entries is 4.5 sigma = 800 entries above the mean)
    const size_t avg_bin = ns / nbins;
    u32 bin_cap = (u32)(((avg_bin + avg_bin / 16 + 512 + 1023) / 1024) * 1024);
    const bool use_bin = avg_bin >= 4096 && bin_cap <= BIN_CAP_MAX && opt().bin_sort;
    TRH_TRY(L.bin_starts.ensure(chunk * Ws * nbins * 4));
    TRH_TRY(L.starts.ensure(chunk * Ws * nb1 * 4));
    TRH_TRY(L.bucket_cnt.ensure(chunk * Ws * nb1 * 4 + 16));
    TRH_TRY(L.ends.ensure(chunk * Ws * nb1 * 4));
    const unsigned range_blocks = (nb1 + RANGE_BLOCK - 1) / RANGE_BLOCK;  // <= 2^17 / 1024 + 1 = 129 < RANGE_BLOCK threads
    TRH_TRY(L.seg_bucket.ensure(chunk * Ws * range_blocks * 4 + 16));  // block totals of the chunked passes' range scan
    TRH_TRY(L.first.ensure(chunk * Ws * nseg0 * sizeof(XYZZzMem)));
    TRH_TRY(L.last.ensure(chunk * Ws * nseg0 * sizeof(XYZZzMem)));
    TRH_TRY(L.direct.ensure(chunk * Ws * nb1 * sizeof(XYZZzMem)));
    TRH_TRY(L.heavy.ensure(chunk * heavy_stride0 * 4 + 16));
    TRH_TRY(L.buckets.ensure(chunk * Ws * nbk * sizeof(XYZZzMem)));
    TRH_TRY(L.partials.ensure(chunk * Ws * rblocks * sizeof(XYZZzMem)));
    if (!bases_z && !fb) TRH_TRY(m.bases_z.ensure(n * ZREC + ZREC));
    const uint4* bz = fb ? (const uint4*)fb->table : bases_z ? (const uint4*)bases_z : m.bases_z.as<uint4>();
    // Lean sort: a caller that vouches for uniformly random scalars (dense_hint: the IPA's rounds) gets the whole-bin LDS sort WITHOUT the four
    // launches of the chunked fallback behind it (they return at once unless a bin overflowed the LDS: 19 us of a round's 680).  The "a bin
    // did overflow" flags travel to the host behind the window sums, and msm_finish repeats the MSM with the fallback when one is set
    // (the scalars must stay as they are until msm_finish: true of every caller that sets dense_hint).
    // (a shape that overflowed before -- k = 16: the 3-bit top window of the c = 14 table puts n / 5 extra entries into each of the buckets 1 .. 4 of
    //  the shared set, every round -- is not tried again: the opening's rounds all have that shape and would each pay the MSM twice)
    const bool lean_sort = m.dense_hint && !m.force_fallback && use_bin && batch <= chunk && (size_t)batch * Ws <= 256 && n != 0 && !(m.lean_off_n == n && m.lean_off_c == cb);
    const size_t flag_off = batch * Ws * sizeof(XYZZMem);
    TRH_TRY(m.window_sums.ensure(flag_off + (lean_sort ? batch * Ws * 4 : 0)));
    const size_t hs = flag_off + (lean_sort ? batch * Ws * 4 : 0);
    if (hs > m.host_sums_cap) {
        if (m.host_sums) (void)hipHostFree(m.host_sums);
        TRH_HIP_TRY(hipHostMalloc(&m.host_sums, hs + 4096, hipHostMallocDefault));
        m.host_sums_cap = hs + 4096;
    }
    if (!(c.attr_done & ATTR_MSM)) {  // per device
        TRH_HIP_TRY(hipFuncSetAttribute((const void*)msm_bin_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_CAP_MAX * 4));
        TRH_HIP_TRY(hipFuncSetAttribute((const void*)msm_partition_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PART_TILE * 4 + 2048 * 12));
        c.attr_done |= ATTR_MSM;
    }
    // unit path (see msm_sparse_emit_kernel): fixed-base mode only (one flat bucket set per item).  Lone commitments ask the sampler too
    // (tools/lone_sparse_probe.py at k = 18, device scalars: a flag column 0.716 -> 0.417 ms; the vote -- one 12 us kernel and a short
    // synchronisation -- costs the others 0.02 - 0.06 ms: word 0.629 -> 0.650, even-bits 0.689 -> 0.723, full-size 0.646 -> 0.710; over the
    // 497 single-call commitments of the literal k = 18 replay 515 -> 460 ms).  The IPA's round MSMs never ask (dense_hint).  The compact pipeline without the unit path loses on a lone commitment (round 4, first half: flag
    // 0.735 vs 0.719 ms, even-bits 0.791 vs 0.691): a lone MSM's time is its latency chain, not the digit slots
    const bool sparse_ok = opt().sparse && fb && !m.dense_hint && !m.no_sparse_vote && n >= 4096 && ns / 8 / SP_LISTS >= 1024 && c.window_override == 0;
    if (sparse_ok) {
        // [counters: chunk x SP_CNT lines][partial sums: chunk x SP_PARTS raw points]
        TRH_TRY(L.sparse.ensure((size_t)chunk * SP_CNT * SP_PAD * 4 + (size_t)chunk * SP_PARTS * sizeof(XYZZzMem) + 64));
        if (!m.sp_host) TRH_HIP_TRY(hipHostMalloc(&m.sp_host, SP_MAX_CHUNK * SP_CNT * 4 + 4 + SP_MAX_CHUNK + 64, hipHostMallocDefault));
    }
    if (m.reserve_only) return TRH_OK;  // trh_bases_reserve: the buffers above are what a launch of this shape needs
    const bool timing = c.timing && batch <= chunk && !sparse_ok;  // one pass over the phases
    if (timing && !m.ev[0]) for (int k = 0; k < 6; ++k) TRH_HIP_TRY(hipEventCreate(&m.ev[k]));

    if (!n) {
        TRH_HIP_TRY(hipMemsetAsync(m.window_sums.p, 0, hs, s));
        if (timing) for (int k = 0; k <= 5; ++k) TRH_HIP_TRY(hipEventRecord(m.ev[k], s));
    }
    // One chunk of items [b0, b0 + nb) from the digits to the window sums.
    //   PIPE_PLAIN    recode into the W x n digit space, entry counts read back for batches (adaptive segments)
    //   PIPE_DENSE    the same for columns the sparse classifier has found full-size (no read-back: the slot count is the entry count)
    //   PIPE_COMPACT  the digits are already there -- msm_sparse_emit_kernel's compact arrays of `cap` slots per item with their level-1
    //                 histogram -- and the sorted entries are remapped to flat table indices before the accumulation
    enum { PIPE_PLAIN, PIPE_DENSE, PIPE_COMPACT };
    const u32 sp_subcap = (u32)(ns / 8 / SP_LISTS), sp_cap = sp_subcap * SP_LISTS;  // a column with more than W n / 8 entries is dense
    auto pipeline = [&](size_t b0, unsigned nb, int mode, size_t entry_sum, u32 entry_most) -> int {
        const bool compact = mode == PIPE_COMPACT;
        const size_t nse = compact ? (size_t)sp_cap : ns;  // slots per item and bucket set
        const uint4* sc = (const uint4*)((const char*)scalars_dev + b0 * stride * 32);
        if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[0], s));
        // small launches (a lone MSM, an IPA round): counts, oversize flags, heavy list and bucket counters cleared by one launch up front instead
        // of four fill kernels along the way (none of them is written before the kernel that the old memset preceded; adaptive batches keep
        // the memsets: their heavy list is sized after the read-back)
        const bool zero_fused = !compact && batch < 8 && mode != PIPE_DENSE;
        if (zero_fused) {
            auto up16 = [](size_t b) { return (u32)((b + 15) / 16); };
            ZeroRanges zr{};
            // range 0 is the whole counts buffer: bin counts, tile flags, oversize flags, entry totals (the buffers below are allocated 16 bytes
            // longer than their contents, so that the 16-byte granules of this kernel never leave them)
            zr.p[0] = (uint4*)L.counts.p; zr.n16[0] = up16((size_t)chunk * Ws * nbins * 4 + flag_bytes + 2 * (size_t)chunk * Ws * 4);
            zr.p[1] = nullptr; zr.n16[1] = 0u;
            zr.p[2] = (uint4*)L.heavy.p; zr.n16[2] = up16((size_t)nb * heavy_stride0 * 4);
            zr.p[3] = (uint4*)L.bucket_cnt.p; zr.n16[3] = up16((size_t)nb * Ws * nb1 * 4);
            const u32 most = zr.n16[3] > zr.n16[0] ? zr.n16[3] : zr.n16[0];
            hipLaunchKernelGGL(msm_zero_ranges_kernel, dim3((most + 255) / 256 < 512 ? (most + 255) / 256 : 512), dim3(256), 0, s, zr);
            TRH_HIP_TRY(hipGetLastError());
        }
        if (!compact) {
            if (!zero_fused) TRH_HIP_TRY(hipMemsetAsync(L.counts.p, 0, fb ? (size_t)chunk * Ws * nbins * 4 + flag_bytes : (size_t)nb * Ws * nbins * 4, s));  // counts (+ the tile flags behind them)
            unsigned gb = (unsigned)((n + 255) / 256);
            if (gb > 2048) gb = 2048;
            hipLaunchKernelGGL((msm_recode_kernel<SF>), dim3(gb, 1, nb), dim3(256), recode_use_lds ? recode_lds : 0, s, sc, n, mont, cb, W,
                               L.digits.as<u32>(), L.counts.as<u32>(), k2, nbins, recode_use_lds, stride, fb ? 1 : 0,
                               tails_dev ? (const uint4*)tails_dev + 2 * b0 : nullptr, tile_flags, 14u, part_tiles);
            static_assert(PART_TILE == 1 << 14, "tile_log of msm_recode_kernel");
        }
        if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[1], s));
        const bool bin_sort = use_bin && !compact;  // compact lists: ~2^11 entries per bin, and a flag column has them all in one: the chunked passes
        if (bin_sort && !zero_fused) TRH_HIP_TRY(hipMemsetAsync(oversize, 0, (size_t)chunk * Ws * 4, s));
        hipLaunchKernelGGL(msm_offsets_kernel, dim3(Ws, 1, nb), dim3(1024), 0, s, L.counts.as<u32>(), L.bin_starts.as<u32>(), nbins, bin_sort ? oversize : nullptr, bin_cap,
                           (adaptive && mode == PIPE_PLAIN) ? totals : nullptr);
        u32 seg_len = seg_len0, nseg = nseg0, heavy_stride = heavy_stride0;
        unsigned heavy_blocks = heavy_blocks0;
        bool resize = false;
        if (mode == PIPE_DENSE) {  // full-size columns: the slot count is the entry count
            seg_len = 128;
            while (seg_len > 16 && (size_t)W * n * nb / seg_len < ((size_t)1 << 19)) seg_len >>= 1;
            nseg = (u32)((ns + seg_len - 1) / seg_len);
            resize = true;
        } else if (compact || (adaptive && (size_t)nb * Ws * 4 <= 4096)) {
            size_t sum = entry_sum;
            u32 most = entry_most;
            if (!compact) {
                u32* ht = (u32*)c.pinned_land;
                TRH_HIP_TRY(hipMemcpyAsync(ht, totals, (size_t)nb * Ws * 4, hipMemcpyDeviceToHost, s));
                TRH_HIP_TRY(hipStreamSynchronize(s));
                sum = 0; most = 0;
                for (size_t q = 0; q < (size_t)nb * Ws; ++q) { sum += ht[q]; most = ht[q] > most ? ht[q] : most; }
            }
            // segments for >= 2^17 live threads (round 4, compact lists: 2^16 2.67 / 1.84 / 2.13 / 2.96 / 2.85 ms for the five sparse batches of the k = 18
            // proof, 2^17 2.37 / 1.85 / 2.03 / 2.88 / 2.88, 2^18 2.34 / 1.71 / 2.06 / 2.98 / 3.05: shorter segments shorten the accumulation's chains and
            // lengthen the combine's)
            constexpr int target_log = 17;
            seg_len = 128;
            while (seg_len > 16 && sum / seg_len < ((size_t)1 << target_log)) seg_len >>= 1;
            nseg = (most + seg_len - 1) / seg_len;  // segments beyond the longest list would find nothing
            if (nseg == 0) nseg = 1;
            resize = true;
        }
        if (resize) {
            const size_t mh = (size_t)Ws * nseg / HEAVY_PIECES + 1;
            heavy_stride = (u32)(mh + 1);
            heavy_blocks = (unsigned)(mh < 256 ? mh : 256);
            TRH_TRY(L.first.ensure(chunk * Ws * nseg * sizeof(XYZZzMem)));
            TRH_TRY(L.last.ensure(chunk * Ws * nseg * sizeof(XYZZzMem)));
            TRH_TRY(L.heavy.ensure(chunk * heavy_stride * 4));
        }
        // the fused zeroing cleared nb * heavy_stride0 words up front: it must not be combined with a resized heavy list
        if (zero_fused && (resize || heavy_stride != heavy_stride0)) { set_error("msm: internal error: fused zeroing with resized segments"); return TRH_EINVAL; }
        if (!zero_fused) TRH_HIP_TRY(hipMemsetAsync(L.heavy.p, 0, (size_t)nb * heavy_stride * 4, s));
        hipLaunchKernelGGL(msm_partition_kernel, dim3((unsigned)((nse + PART_TILE - 1) / PART_TILE), Ws, nb), dim3(PART_THREADS), (size_t)PART_TILE * 4 + (size_t)nbins * 12, s,
                           L.digits.as<u32>(), L.counts.as<u32>(), L.parted.as<u32>(), nse, k2, nbins, idx_bits, compact ? (const unsigned char*)nullptr : tile_flags);
        {
            const dim3 cgrid((unsigned)((nse + BS_CHUNK - 1) / BS_CHUNK), Ws, nb);
            const u32* gate = bin_sort ? oversize : nullptr;  // the chunked passes return at once when the bin sort did the work
            if (bin_sort)
                hipLaunchKernelGGL(msm_bin_sort_kernel, dim3(nbins, Ws, nb), dim3(BIN_THREADS), (size_t)bin_cap * 4, s, L.parted.as<u32>(), L.bin_starts.as<u32>(), L.counts.as<u32>(),
                                   L.sorted.as<u32>(), L.starts.as<u32>(), L.ends.as<u32>(), nse, k2, nbins, idx_bits, nbk, oversize);
            if (!(lean_sort && bin_sort)) {
            if (!zero_fused) TRH_HIP_TRY(hipMemsetAsync(L.bucket_cnt.p, 0, (size_t)nb * Ws * nb1 * 4, s));
            hipLaunchKernelGGL((msm_bucket_pass_kernel<false>), cgrid, dim3(BS_THREADS), 0, s, L.parted.as<u32>(), L.bin_starts.as<u32>(), L.counts.as<u32>(),
                               L.bucket_cnt.as<u32>(), L.sorted.as<u32>(), nse, k2, nbins, idx_bits, nbk, gate);
            // block totals live in seg_bucket, which is only filled afterwards
            hipLaunchKernelGGL(msm_bucket_block_sums_kernel, dim3(range_blocks, Ws, nb), dim3(RANGE_BLOCK), 0, s, L.bucket_cnt.as<u32>(), L.seg_bucket.as<u32>(), nbk, gate);
            hipLaunchKernelGGL(msm_bucket_ranges_kernel, dim3(range_blocks, Ws, nb), dim3(RANGE_BLOCK), 0, s, L.bucket_cnt.as<u32>(), L.seg_bucket.as<u32>(), L.starts.as<u32>(),
                               L.ends.as<u32>(), nbk, gate);
            hipLaunchKernelGGL((msm_bucket_pass_kernel<true>), cgrid, dim3(BS_THREADS), 0, s, L.parted.as<u32>(), L.bin_starts.as<u32>(), L.counts.as<u32>(),
                               L.bucket_cnt.as<u32>(), L.sorted.as<u32>(), nse, k2, nbins, idx_bits, nbk, gate);
            }
            if (compact) {
                unsigned gr = (entry_most + 255) / 256;
                if (gr > 512) gr = 512;
                hipLaunchKernelGGL(msm_sparse_remap_kernel, dim3(gr ? gr : 1, nb), dim3(256), 0, s, L.sorted.as<u32>(), L.digits.as<u32>() + (size_t)chunk * sp_cap, L.ends.as<u32>(), nse, nbk);
            }
        }
        if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[2], s));
        if (b0 == 0 && !bases_z && !fb)
            hipLaunchKernelGGL((msm_convert_bases_kernel<BF>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint4*)bases_dev, m.bases_z.as<uint4>(), n);
        hipLaunchKernelGGL((msm_accumulate_seg_kernel<BF>), dim3((nseg + 255) / 256, Ws, nb), dim3(256), 0, s, bz, L.sorted.as<u32>(),
                           L.ends.as<u32>(), L.first.as<XYZZzMem>(), L.last.as<XYZZzMem>(), L.direct.as<XYZZzMem>(), nse, nbk, nseg, seg_len,
                           (lean_sort && bin_sort) ? oversize : (const u32*)nullptr);
        if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[5], s));
        {
            const size_t pieces = compact ? 4 : ns / ((size_t)nbk * seg_len);  // expected pieces per bucket
#define TRH_LAUNCH_COMBINE(G)                                                                                                                          \
    hipLaunchKernelGGL((msm_combine_kernel<BF, G>), dim3((unsigned)(((size_t)nbk * G + 255) / 256), Ws, nb), dim3(256), 0, s, L.starts.as<u32>(), L.ends.as<u32>(), \
                       L.first.as<XYZZzMem>(), L.last.as<XYZZzMem>(), L.direct.as<XYZZzMem>(), L.buckets.as<XYZZzMem>(), nbk, nseg, seg_len, L.heavy.as<u32>(), heavy_stride, \
                       (lean_sort && bin_sort) ? oversize : (const u32*)nullptr)
            // measured on the IPA opening at k = 18 (32 pieces per bucket): 24.0 ms with one lane per bucket, 23.0 with 4, 26.1 with 16
            // (idle lanes of the wider groups still occupy the SIMD)
            // sparse lists cut into short segments (adaptive): the few buckets that hold anything hold many pieces (an even-bits word column:
            // 256 buckets of 512 entries = 32 pieces each, added one after the other by one thread: 1.6 ms per batch) while most groups find an
            // empty bucket and leave at once -- a quarter wave per bucket
            // (16 lanes per bucket for these: 2^15 buckets x 64 columns x 16 lanes of mostly empty groups cost more than the chains: 3.4 -> 5.4 ms)
            // quad form only where the caller vouches for uniformly full-size scalars (dense_hint: the IPA's rounds, three or four pieces per bucket):
            // opening 12.8 -> 12.4 ms; on a lone even-bits column (256 buckets of 32 pieces, which four lanes share better than one quad walks)
            // it measured 0.52 -> 0.58 ms, full-size and word columns the same
            if (pieces >= 3 && nbk >= 4 && !compact && opt().reduce_q4 && m.dense_hint && (size_t)Ws * nb <= 8)
                hipLaunchKernelGGL((msm_combine_q4_kernel<BF>), dim3((unsigned)(((size_t)nbk * 4 + 255) / 256), Ws, nb), dim3(256), 0, s, L.starts.as<u32>(), L.ends.as<u32>(),
                                   L.first.as<XYZZzMem>(), L.last.as<XYZZzMem>(), L.direct.as<XYZZzMem>(), L.buckets.as<XYZZzMem>(), nbk, nseg, seg_len, L.heavy.as<u32>(), heavy_stride,
                                   (lean_sort && bin_sort) ? oversize : (const u32*)nullptr);
            else if (pieces >= 3 && nbk >= 4) TRH_LAUNCH_COMBINE(4);
            else TRH_LAUNCH_COMBINE(1);
#undef TRH_LAUNCH_COMBINE
        }
        hipLaunchKernelGGL((msm_combine_heavy_kernel<BF>), dim3(heavy_blocks, 1, nb), dim3(256), 0, s, L.starts.as<u32>(), L.ends.as<u32>(), L.first.as<XYZZzMem>(),
                           L.last.as<XYZZzMem>(), L.buckets.as<XYZZzMem>(), nbk, nseg, seg_len, L.heavy.as<u32>(), (u32)Ws, heavy_stride);
        if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[3], s));
        {
            // compact lists keep the geometry of the plain ones (round 4: shorter slices made their reduction SLOWER -- 1024 threads per item
            // 0.37 - 0.64 ms per batch of 64, 8192: 0.45 - 1.38: it is bound by its 2 x 2^15 x 64 point additions, not by its depth)
            const u32 r_tpw = tpw, r_slice = slice, r_blocks = rblocks;
            // quad-lane form (curve_q4.h) when the launch is ONE latency chain -- at most 8 bucket sets (a lone fixed-base commitment, an IPA
            // round, a small fixed-base batch) and four lanes per slice within 2^16 lanes (one wave per SIMD).  Measured
            // (profiles/r05_q4_reduce_ab.txt): IPA k = 18 14.9 -> 13.0 ms, a lone 2^18 commitment 0.71 -> 0.61 ms; with 16 windows the quads
            // crowd each other out (2^24: reduction 0.43 -> 0.52 ms) and 2^17 lanes lose to 2^16 (IPA 13.9 against 13.0 ms).  Option
            // reduce_q4 = 0 keeps the one-thread-per-slice kernels everywhere
            const int q4_knob = opt().reduce_q4;
            constexpr int q4_lanes_log = 16, q4_sets = 8;  // (16 - 32 sets / 2^17 - 2^19 lanes lose at 2^18 .. 2^22: profiles/r06_q4_small_raw.txt; 2^15 / 2^14 lanes: opening 8.2 -> 8.6 ms)
            u32 q4_tpw = r_tpw;
            while (q4_tpw > 1024 && (size_t)q4_tpw * 4 * Ws * nb > ((size_t)1 << q4_lanes_log)) q4_tpw >>= 1;
            if (q4_knob && !compact && (size_t)Ws * nb <= (size_t)q4_sets && q4_tpw >= 64 && (size_t)q4_tpw * 4 * Ws * nb <= ((size_t)1 << q4_lanes_log) && nbk % q4_tpw == 0) {
                const u32 q4_blocks = (q4_tpw + 63) / 64;
                TRH_TRY(L.partials.ensure((size_t)chunk * Ws * q4_blocks * sizeof(XYZZzMem)));
                hipLaunchKernelGGL((msm_reduce_q4_kernel<BF>), dim3(q4_blocks, Ws, nb), dim3(256), 0, s, L.buckets.as<XYZZzMem>(), L.partials.as<XYZZzMem>(), nbk, nbk / q4_tpw, q4_tpw);
                hipLaunchKernelGGL((msm_window_sum_q4_kernel<BF>), dim3(Ws, 1, nb), dim3(256), 0, s, L.partials.as<XYZZzMem>(), m.window_sums.as<XYZZMem>() + b0 * Ws, q4_blocks,
                                   lean_sort ? oversize : (const u32*)nullptr, (u32*)((char*)m.window_sums.p + flag_off), (u32)(nb * Ws));
            } else {
            hipLaunchKernelGGL((msm_reduce_kernel<BF>), dim3(r_blocks, Ws, nb), dim3(256), 0, s, L.buckets.as<XYZZzMem>(), L.partials.as<XYZZzMem>(), nbk, r_slice, r_tpw);
            hipLaunchKernelGGL((msm_window_sum_kernel<BF>), dim3(Ws, 1, nb), dim3(256), 0, s, L.partials.as<XYZZzMem>(), m.window_sums.as<XYZZMem>() + b0 * Ws, r_blocks,
                               lean_sort ? oversize : (const u32*)nullptr, (u32*)((char*)m.window_sums.p + flag_off), (u32)(nb * Ws));
            }
        }
        if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[4], s));
        return TRH_OK;
    };

    // Fixed-base mode (the commitments of create_proof): the sparse columns of a chunk take the compact pipeline in one pass over the whole
    // chunk (the dense columns' lists are empty there), the dense ones the plain pipeline in runs of consecutive items, which overwrite
    // their window sums.  One synchronisation per chunk either way (the plain path's adaptive read-back is not needed).
    auto sparse_chunk = [&](size_t b0, unsigned nb) -> int {
        const uint4* sc = (const uint4*)((const char*)scalars_dev + b0 * stride * 32);
        const uint4* tl = tails_dev ? (const uint4*)tails_dev + 2 * b0 : nullptr;
        u32* const sp_count = L.sparse.as<u32>();
        XYZZzMem* const part = (XYZZzMem*)(sp_count + (size_t)chunk * SP_CNT * SP_PAD);
        u32* const digits = L.digits.as<u32>();                       // [chunk][cap] compact digits, [chunk][cap] flat table indices, [chunk][cap] unit entries
        u32* const flat = digits + (size_t)chunk * sp_cap;            // (3 x chunk x W n / 8 x 4 B: three eighths of the buffer)
        u32* const units = flat + (size_t)chunk * sp_cap;
        u32* const hc = (u32*)m.sp_host;                       // [nb][SP_CNT] counters, the chunk's votes, then nb class bytes
        u32* const hvotes = hc + SP_MAX_CHUNK * SP_CNT;
        unsigned char* const hd = (unsigned char*)(hvotes + 1);
        TRH_HIP_TRY(hipMemsetAsync(sp_count, 0, (size_t)nb * SP_CNT * SP_PAD * 4, s));
        hipLaunchKernelGGL((msm_sparse_sample_kernel<SF>), dim3(nb), dim3(256), 0, s, sc, n, mont, cb, W, stride, tl, sp_count, 2 * sp_cap);
        // Measured on the k = 18 proof, chunk by chunk on one box: the compact pipeline alone is no faster than the plain one (word columns
        // 2.93 against 2.55 ms, sorted / even-bits 2.92 against 3.08, mixed 2.42 against 2.37); what pays is the unit path (flag chunks 0.64 /
        // 0.96 against 1.64 / 1.82 ms).  So the sampler's vote is read first (one short synchronisation) and a chunk that is not flag-like
        // goes through the plain pipeline as if the path did not exist.
        TRH_HIP_TRY(hipMemcpyAsync(hvotes, sp_count + SP_VOTES, 4, hipMemcpyDeviceToHost, s));
        TRH_HIP_TRY(hipStreamSynchronize(s));
        if (*hvotes != nb) return pipeline(b0, nb, PIPE_PLAIN, 0, 0);
        TRH_HIP_TRY(hipMemsetAsync(digits, 0, (size_t)nb * sp_cap * 4, s));
        TRH_HIP_TRY(hipMemsetAsync(L.counts.p, 0, (size_t)nb * nbins * 4, s));
        unsigned gbe = (unsigned)((n + 255) / 256);
        if (gbe > 1024) gbe = 1024;
        hipLaunchKernelGGL((msm_sparse_emit_kernel<SF>), dim3(gbe, 1, nb), dim3(256), 0, s, sc, n, mont, cb, W, stride, tl, sp_count, digits, flat, units, sp_subcap, L.counts.as<u32>(), k2, nbins);
        // unit path: the lists (and the digit entries of tiny columns) are summed while the host reads the counters; the kernel reads the
        // chunk's vote and classifies the columns by the same rule as the host below
        hipLaunchKernelGGL((msm_unit_sum_kernel<BF>), dim3(SP_PARTS, nb), dim3(256), 0, s, bz, sp_count, units, digits, flat, sp_subcap, part);
        TRH_HIP_TRY(hipMemcpy2DAsync(hc, 4, sp_count, SP_PAD * 4, 4, (size_t)nb * SP_CNT, hipMemcpyDeviceToHost, s));
        TRH_HIP_TRY(hipMemcpyAsync(hvotes, sp_count + SP_VOTES, 4, hipMemcpyDeviceToHost, s));
        TRH_HIP_TRY(hipStreamSynchronize(s));
        const int unit_mode = *hvotes == nb;
        size_t sum = 0;
        u32 most = 0, n_general = 0, n_neutral = 0, n_dense = 0;
        for (unsigned z = 0; z < nb; ++z) {
            const int cls = sp_classify(hc + (size_t)z * SP_CNT, 1, sp_subcap, unit_mode);
            hd[z] = (unsigned char)cls;
            size_t cz = 0;
            for (int g = 0; g < SP_LISTS; ++g) cz += hc[(size_t)z * SP_CNT + g];
            if (cls == SP_CLASS_DENSE) {
                ++n_dense;
                if (hc[(size_t)z * SP_CNT] != SP_DENSE) ++n_neutral;  // it emitted until a list ran full
            } else if (cls == SP_CLASS_TINY) {
                if (cz) ++n_neutral;
            } else {
                ++n_general; sum += cz; most = cz > most ? (u32)cz : most;
            }
        }
        const bool run_compact = n_general != 0 && most != 0;
        if (run_compact) {
            if (n_neutral)  // what the columns that stay out of the compact pipeline appended must not be sorted
                hipLaunchKernelGGL(msm_sparse_neutralise_kernel, dim3(64, nb), dim3(256), 0, s, sp_count, sp_subcap, digits, (size_t)sp_cap, L.counts.as<u32>(), nbins);
            TRH_TRY(pipeline(b0, nb, PIPE_COMPACT, sum, most));
        }
        if (n_dense < nb && (unit_mode || !run_compact))
            hipLaunchKernelGGL((msm_unit_final_kernel<BF>), dim3(nb), dim3(64), 0, s, sp_count, sp_subcap, part, m.window_sums.as<XYZZMem>() + b0, run_compact ? 1 : 0);
        for (unsigned z = 0; z < nb;) {  // the dense columns, in runs
            if (hd[z] != SP_CLASS_DENSE) { ++z; continue; }
            unsigned e = z + 1;
            while (e < nb && hd[e] == SP_CLASS_DENSE) ++e;
            TRH_TRY(pipeline(b0 + z, e - z, PIPE_DENSE, 0, 0));
            z = e;
        }
        return TRH_OK;
    };
    for (size_t b0 = 0; n && b0 < batch; b0 += chunk) {
        const unsigned nb = (unsigned)(b0 + chunk <= batch ? chunk : batch - b0);
        if (sparse_ok) TRH_TRY(sparse_chunk(b0, nb));
        else TRH_TRY(pipeline(b0, nb, PIPE_PLAIN, 0, 0));
    }
    TRH_HIP_TRY(hipGetLastError());
    TRH_HIP_TRY(hipMemcpyAsync(m.host_sums, m.window_sums.p, hs, hipMemcpyDeviceToHost, s));
    m.pending_curve = BF::ID;
    m.pending_windows = Ws;
    m.pending_c = cb;
    m.pending_batch = batch;
    m.pending_stream = s;
    m.pending_owner = nullptr;
    m.ev_valid = timing;
    m.lean_pending = lean_sort;
    if (lean_sort) {  // what msm_finish needs to run this MSM again with the chunked passes
        m.retry.bases_dev = bases_dev; m.retry.bases_z = bases_z; m.retry.scalars_dev = scalars_dev; m.retry.tails_dev = tails_dev;
        m.retry.n = n; m.retry.batch = batch; m.retry.stride = stride; m.retry.mont = mont;
        m.retry.has_fb = fb != nullptr;
        if (fb) m.retry.fb = *fb;
    }
    return TRH_OK;
}

// host: Horner over windows, normalise
template <class BF>
void combine_windows_host(const XYZZMem* ws, int W, int cb, u64* out_xyz) {
    // 4 x 64-bit limbs on the CPU (hostcombine.h): the nine-limb form of the device code costs 0.26 ms per MSM here, this 0.06
    hostcombine::combine_windows<BF>((const uint64_t*)ws, W, cb, (uint64_t*)out_xyz);
}

template <class BF>
int msm_finish_t(hipStream_t s, u64* out_xyz, size_t batch) {
    Ctx& c = ctx();
    MsmScratch& m = c.msm;
    if (m.pending_curve != BF::ID || m.pending_batch != batch || m.pending_stream != s) { set_error("msm_finish: no matching MSM enqueued on this context and stream"); return TRH_EINVAL; }
    const int curve_id = m.pending_curve;
    m.pending_curve = -1;  // whatever happens below, the context is free for the next MSM
    TRH_HIP_TRY(hipStreamSynchronize(s));
    if (m.lean_pending) {  // the lean sort's promise: no bin overflowed the LDS of msm_bin_sort_kernel.  Otherwise: once more, with the chunked passes
        m.lean_pending = false;
        const u32* fl = (const u32*)((const char*)m.host_sums + batch * m.pending_windows * sizeof(XYZZMem));
        bool over = false;
        for (size_t q = 0; q < batch * (size_t)m.pending_windows; ++q) over = over || fl[q] != 0;
        if (over) {
            const bool hint = m.dense_hint;
            m.force_fallback = true; m.dense_hint = true;
            const int rc = msm_enqueue(curve_id, m.retry.bases_dev, m.retry.bases_z, m.retry.scalars_dev, m.retry.n, m.retry.batch, m.retry.stride, m.retry.mont, s,
                                       m.retry.has_fb ? &m.retry.fb : nullptr, m.retry.tails_dev);
            m.force_fallback = false; m.dense_hint = hint;
            m.pending_curve = -1;
            TRH_TRY(rc);
            TRH_HIP_TRY(hipStreamSynchronize(s));
            ++m.lean_retries;
            m.lean_off_n = m.retry.n; m.lean_off_c = m.pending_c;
        }
    }
    const XYZZMem* ws = (const XYZZMem*)m.host_sums;
    if (m.pending_windows > 1 && batch >= 2 && batch <= 256 && !c.helper && !c.helper_failed) {
        try { c.helper = new HostHelper(); } catch (...) { c.helper_failed = true; }  // no thread to be had: the Horners run one after the other below
    }
    if (m.pending_windows > 1 && batch >= 2 && batch <= 256 && c.helper) {
        // Horner over the windows is ~250 doublings per result (70 us) and the results are independent: the upper half of the batch on the context's
        // helper thread (hosthelper.h), the lower half here, then one inversion for all of them
        const int W = m.pending_windows, cb = m.pending_c;
        hostcombine::P acc[256];
        const size_t mid = batch / 2;
        const uint64_t* w64 = (const uint64_t*)ws;
        c.helper->start([&acc, w64, W, cb, mid, batch] { for (size_t i = mid; i < batch; ++i) acc[i] = hostcombine::horner<BF>(w64 + i * (size_t)W * 16, W, cb); });
        for (size_t i = 0; i < mid; ++i) acc[i] = hostcombine::horner<BF>(w64 + i * (size_t)W * 16, W, cb);
        c.helper->wait();
        hostcombine::normalise_batch<BF>(acc, batch, (uint64_t*)out_xyz);
    } else
    hostcombine::combine_windows_batch<BF>((const uint64_t*)ws, m.pending_windows, m.pending_c, batch, (uint64_t*)out_xyz);  // one inversion for the whole batch
    if (m.ev_valid) {
        float t01, t12, t23, t34, tt, t25;
        TRH_HIP_TRY(hipEventElapsedTime(&t25, m.ev[2], m.ev[5]));
        c.last.accumulate_kernel_ms = t25;
        TRH_HIP_TRY(hipEventElapsedTime(&t01, m.ev[0], m.ev[1]));
        TRH_HIP_TRY(hipEventElapsedTime(&t12, m.ev[1], m.ev[2]));
        TRH_HIP_TRY(hipEventElapsedTime(&t23, m.ev[2], m.ev[3]));
        TRH_HIP_TRY(hipEventElapsedTime(&t34, m.ev[3], m.ev[4]));
        TRH_HIP_TRY(hipEventElapsedTime(&tt, m.ev[0], m.ev[4]));
        c.last.total_ms = tt; c.last.digits_ms = t01; c.last.sort_ms = t12; c.last.accumulate_ms = t23; c.last.reduce_ms = t34;
    }
    else if (c.timing) {
        // timing was asked for but this MSM was not timed phase by phase (a tabled set through the sampler, a batch beyond one chunk):
        // zeros, not the previous MSM's figures
        c.last.total_ms = c.last.digits_ms = c.last.sort_ms = c.last.accumulate_ms = c.last.reduce_ms = c.last.accumulate_kernel_ms = 0;
    }
    c.last.window_bits = m.pending_c;
    c.last.windows = m.pending_windows;
    if (m.tile_sum_valid && !m.in_tile) {  // last tile of a tiled MSM: add the earlier tiles
        u64 two[24];
        memcpy(two, m.tile_sum, 96);
        memcpy(two + 12, out_xyz, 96);
        m.tile_sum_valid = false;
        return point_sum_host_t<BF>(two, 2, out_xyz);
    }
    return TRH_OK;
}

template <class BF>
int point_sum_host_t(const u64* pts, size_t count, u64* out) {
    XYZZ<BF> acc = xyzz_identity<BF>();
    for (size_t i = 0; i < count; ++i) {
        JacobianMem j;
        memcpy(&j, pts + 12 * i, 96);
        acc = xyzz_add(acc, xyzz_from_jacobian(jac_load<BF>(j)));
    }
    JacobianMem r;
    jac_store(jac_from_affine(xyzz_to_affine(acc)), r);
    memcpy(out, &r, 96);
    return TRH_OK;
}

}  // namespace

size_t msm_small_max_pairs() { return SMALL_MAX_N; }
int msm_fixed_base_windows(int c) { return num_windows(c); }
// the partitioned entries pack 7 low bucket bits above the flat table index
bool msm_fixed_base_fits(size_t n, int c) {
    if (c < 2 || c > MAX_C_FIXED) return false;
    return (size_t)num_windows(c) * n <= ((size_t)1 << 24);
}
int msm_build_table(int curve, const void* bases_dev, size_t n, int c, void* table_dev, hipStream_t s) {
    if (!n) return TRH_OK;
    const int W = num_windows(c);
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (curve == TRH_PALLAS) hipLaunchKernelGGL((msm_table_kernel<FpParams>), dim3(gb), dim3(256), 0, s, (const uint4*)bases_dev, (uint4*)table_dev, n, c, W);
    else hipLaunchKernelGGL((msm_table_kernel<FqParams>), dim3(gb), dim3(256), 0, s, (const uint4*)bases_dev, (uint4*)table_dev, n, c, W);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}
int msm_enqueue(int curve, const void* bases_dev, const void* bases_z, const void* scalars_dev, size_t n, size_t batch, size_t stride, int mont, hipStream_t s, const MsmFixedBase* fb,
                const void* tails_dev) {
    // pallas: base Fp, scalar Fq; vesta: base Fq, scalar Fp
    if (curve == TRH_PALLAS) return msm_enqueue_t<FqParams, FpParams>(bases_dev, bases_z, scalars_dev, n, batch, stride, mont, s, fb, tails_dev);
    return msm_enqueue_t<FpParams, FqParams>(bases_dev, bases_z, scalars_dev, n, batch, stride, mont, s, fb, tails_dev);
}
int msm_convert_bases(int curve, const void* in_dev, void* out_dev, size_t n, hipStream_t s) {
    if (!n) return TRH_OK;
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (curve == TRH_PALLAS) hipLaunchKernelGGL((msm_convert_bases_kernel<FpParams>), dim3(gb), dim3(256), 0, s, (const uint4*)in_dev, (uint4*)out_dev, n);
    else hipLaunchKernelGGL((msm_convert_bases_kernel<FqParams>), dim3(gb), dim3(256), 0, s, (const uint4*)in_dev, (uint4*)out_dev, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}
int msm_finish(int curve, hipStream_t s, u64* out_xyz, size_t batch) {
    if (curve == TRH_PALLAS) return msm_finish_t<FpParams>(s, out_xyz, batch);
    return msm_finish_t<FqParams>(s, out_xyz, batch);
}
int point_sum_host(int curve, const u64* pts, size_t count, u64* out) {
    if (curve == TRH_PALLAS) return point_sum_host_t<FpParams>(pts, count, out);
    return point_sum_host_t<FqParams>(pts, count, out);
}
int bases_generate_device(int curve, u64 s0, u64 d, u64 first, size_t n, void* out_dev, hipStream_t s) {
    if (!n) return TRH_OK;
    const unsigned gb = (unsigned)((n + 255) / 256);
    if (curve == TRH_PALLAS) hipLaunchKernelGGL((bases_generate_kernel<FpParams>), dim3(gb), dim3(256), 0, s, s0, d, first, n, (uint4*)out_dev);
    else hipLaunchKernelGGL((bases_generate_kernel<FqParams>), dim3(gb), dim3(256), 0, s, s0, d, first, n, (uint4*)out_dev);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}
void msm_release() {
    MsmScratch& m = ctx().msm;
    m.scalars.release(); m.tails.release(); m.bases_z.release(); m.window_sums.release();
    MsmLane& L = m.lane;
    L.digits.release(); L.parted.release(); L.sorted.release(); L.counts.release(); L.bin_starts.release(); L.starts.release(); L.ends.release(); L.bucket_cnt.release();
    L.seg_bucket.release(); L.first.release(); L.last.release(); L.direct.release(); L.heavy.release(); L.buckets.release(); L.partials.release(); L.sparse.release();
    if (m.sp_host) (void)hipHostFree(m.sp_host);
    m.sp_host = nullptr;
    if (m.host_sums) (void)hipHostFree(m.host_sums);
    m.host_sums = nullptr; m.host_sums_cap = 0;
    for (int k = 0; k < 6; ++k) if (m.ev[k]) { (void)hipEventDestroy(m.ev[k]); m.ev[k] = nullptr; }
}

}  // namespace trh
