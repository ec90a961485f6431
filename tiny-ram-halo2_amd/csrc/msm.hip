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
//   4. accumulate one thread per (window, bucket): XYZZ accumulator += affine base (8M + 2S),
//                coalesced 16-byte loads of the 64-byte base, next base prefetched
//   5. reduce    sum_b b * B_b per window: per-thread running sums over a slice of buckets,
//                slice offset by a short double-and-add, LDS tree across the workgroup
//   6. combine   per-window sums -> host (W x 128 B), Horner over windows on the host
//
// Integer work, no MFMA.  Algorithmic HBM bytes: 32 B scalar + 64 B base per pair.
#include <string.h>

#include "ctx.h"

namespace trh {

namespace {

constexpr int MAX_C = 16;
constexpr u32 SIGN_BIT = 0x80000000u;

inline int ilog2_floor(size_t n) {
    int l = 0;
    while ((n >> (l + 1)) != 0) ++l;
    return l;
}

inline int choose_window_bits(size_t n) {
    int o = ctx().window_override;
    if (o >= 2 && o <= MAX_C) return o;
    int c = ilog2_floor(n ? n : 1) / 2 + 4;
    if (c > MAX_C) c = MAX_C;
    if (c < 2) c = 2;
    return c;
}
inline int num_windows(int c) { return 255 / c + 1; }

// ---------------------------------------------------------------------------------------
// 1. recode
// ---------------------------------------------------------------------------------------
template <class SF>
__global__ void __launch_bounds__(256) msm_recode_kernel(const uint4* __restrict__ scalars, size_t n, int mont, int c, int W,
                                                         u32* __restrict__ digits, u32* __restrict__ counts, u32 nb1) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 lo = scalars[2 * i], hi = scalars[2 * i + 1];
    Fe<SF> s;
    s.l[0] = lo.x; s.l[1] = lo.y; s.l[2] = lo.z; s.l[3] = lo.w;
    s.l[4] = hi.x; s.l[5] = hi.y; s.l[6] = hi.z; s.l[7] = hi.w;
    if (mont) s = fe_from_mont(s);
    const u32 mask = (1u << c) - 1u, half = 1u << (c - 1);
    u32 carry = 0;
    for (int j = 0; j < W; ++j) {
        u32 raw = (s.l[0] & mask) + carry;
        // shift the 256-bit value right by c (static register indexing)
#pragma unroll
        for (int k = 0; k < 7; ++k) s.l[k] = (s.l[k] >> c) | (s.l[k + 1] << (32 - c));
        s.l[7] >>= c;
        u32 entry = 0;
        if (raw > half) {  // negative digit raw - 2^c, borrow from the next window
            u32 b = (1u << c) - raw;
            carry = 1;
            entry = b | SIGN_BIT;
            if (b) atomicAdd(&counts[(size_t)j * nb1 + b], 1u);
            else entry = 0;  // raw == 2^c: digit 0 with carry
        } else {
            carry = 0;
            entry = raw;
            if (raw) atomicAdd(&counts[(size_t)j * nb1 + raw], 1u);
        }
        digits[(size_t)j * n + i] = entry;
    }
}

// ---------------------------------------------------------------------------------------
// 2. offsets: per window exclusive scan of counts[0..nb1) -> starts; counts becomes the cursor
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) msm_offsets_kernel(u32* __restrict__ counts, u32* __restrict__ starts, u32 nb1) {
    __shared__ u32 part[1024];
    const int j = blockIdx.x, t = threadIdx.x;
    u32* cnt = counts + (size_t)j * nb1;
    u32* st = starts + (size_t)j * nb1;
    const u32 per = (nb1 + 1023u) / 1024u;
    const u32 lo = t * per, hi = (lo + per < nb1) ? lo + per : nb1;
    u32 sum = 0;
    for (u32 b = lo; b < hi; ++b) sum += cnt[b];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        u32 v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u32 run = part[t] - sum;
    for (u32 b = lo; b < hi; ++b) {
        u32 cv = cnt[b];
        st[b] = run;
        cnt[b] = run;  // cursor starts at the bucket start
        run += cv;
    }
}

// ---------------------------------------------------------------------------------------
// 3. scatter
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) msm_scatter_kernel(const u32* __restrict__ digits, u32* __restrict__ cursor,
                                                          u32* __restrict__ sorted, size_t n, u32 nb1) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= n) return;
    u32 e = digits[(size_t)j * n + i];
    u32 b = e & ~SIGN_BIT;
    if (!b) return;
    u32 pos = atomicAdd(&cursor[(size_t)j * nb1 + b], 1u);
    sorted[(size_t)j * n + pos] = (u32)i | (e & SIGN_BIT);
}

// ---------------------------------------------------------------------------------------
// 4. accumulate
// ---------------------------------------------------------------------------------------
template <class BF>
__device__ __forceinline__ Affine<BF> load_affine(const uint4* __restrict__ bases, u32 idx) {
    const uint4* p = bases + (size_t)idx * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    Affine<BF> r;
    r.x.l[0] = a.x; r.x.l[1] = a.y; r.x.l[2] = a.z; r.x.l[3] = a.w;
    r.x.l[4] = b.x; r.x.l[5] = b.y; r.x.l[6] = b.z; r.x.l[7] = b.w;
    r.y.l[0] = c.x; r.y.l[1] = c.y; r.y.l[2] = c.z; r.y.l[3] = c.w;
    r.y.l[4] = d.x; r.y.l[5] = d.y; r.y.l[6] = d.z; r.y.l[7] = d.w;
    return r;
}

template <class BF>
__device__ __forceinline__ void store_xyzz(XYZZ<BF>* dst, const XYZZ<BF>& v) {
    uint4* p = (uint4*)dst;
    const u32* w = (const u32*)&v;
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
template <class BF>
__device__ __forceinline__ XYZZ<BF> load_xyzz(const XYZZ<BF>* src) {
    const uint4* p = (const uint4*)src;
    XYZZ<BF> v;
    u32* w = (u32*)&v;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint4 q = p[k];
        w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
    }
    return v;
}

template <class BF>
__global__ void __launch_bounds__(256) msm_accumulate_kernel(const uint4* __restrict__ bases, const u32* __restrict__ sorted,
                                                             const u32* __restrict__ starts, const u32* __restrict__ ends,
                                                             XYZZ<BF>* __restrict__ buckets, size_t n, u32 nbk) {
    // thread -> bucket id 1..nbk of window blockIdx.y
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y;
    if (b > nbk) return;
    const u32 nb1 = nbk + 1;
    const u32 lo = starts[(size_t)j * nb1 + b], hi = ends[(size_t)j * nb1 + b];
    const u32* lst = sorted + (size_t)j * n;
    XYZZ<BF> acc = xyzz_identity<BF>();
    if (lo < hi) {
        u32 e = lst[lo];
        Affine<BF> nxt = load_affine<BF>(bases, e & ~SIGN_BIT);
        for (u32 k = lo; k < hi; ++k) {
            Affine<BF> cur = nxt;
            const u32 ce = e;
            if (k + 1 < hi) {  // prefetch the next base while this add runs
                e = lst[k + 1];
                nxt = load_affine<BF>(bases, e & ~SIGN_BIT);
            }
            if (ce & SIGN_BIT) cur.y = fe_neg(cur.y);
            xyzz_madd(acc, cur);
        }
    }
    store_xyzz(&buckets[(size_t)j * nbk + (b - 1)], acc);
}

// ---------------------------------------------------------------------------------------
// 5. reduce: window sum = sum_{b=1..nbk} b * B_b
//    thread t of a window owns buckets t*m+1 .. (t+1)*m; blocks of 256 threads tree-add in LDS.
// ---------------------------------------------------------------------------------------
template <class BF>
__global__ void __launch_bounds__(256) msm_reduce_kernel(const XYZZ<BF>* __restrict__ buckets, XYZZ<BF>* __restrict__ partials,
                                                         u32 nbk, u32 m, u32 threads_per_window) {
    __shared__ XYZZ<BF> sh[256];
    const int j = blockIdx.y;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ<BF> total = xyzz_identity<BF>();
    if (t < threads_per_window) {
        const XYZZ<BF>* bk = buckets + (size_t)j * nbk + (size_t)t * m;
        XYZZ<BF> run = xyzz_identity<BF>(), acc = xyzz_identity<BF>();
        for (int k = (int)m - 1; k >= 0; --k) {
            XYZZ<BF> v = load_xyzz(&bk[k]);
            run = xyzz_add(run, v);
            acc = xyzz_add(acc, run);
        }
        // acc = sum (k+1) * B[k]; the slice starts at global bucket id t*m + 1 -> add (t*m) * run
        const u32 off = t * m;
        if (off) {
            XYZZ<BF> sc = xyzz_identity<BF>();
            int top = 31 - __clz(off);
            for (int i = top; i >= 0; --i) {
                sc = xyzz_dbl(sc);
                if ((off >> i) & 1u) sc = xyzz_add(sc, run);
            }
            acc = xyzz_add(acc, sc);
        }
        total = acc;
    }
    sh[threadIdx.x] = total;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = xyzz_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) store_xyzz(&partials[(size_t)j * gridDim.x + blockIdx.x], sh[0]);
}

// one block per window: sum `count` partials
template <class BF>
__global__ void __launch_bounds__(256) msm_window_sum_kernel(const XYZZ<BF>* __restrict__ partials, XYZZ<BF>* __restrict__ window_sums, u32 count) {
    __shared__ XYZZ<BF> sh[256];
    const int j = blockIdx.x;
    XYZZ<BF> v = xyzz_identity<BF>();
    for (u32 k = threadIdx.x; k < count; k += 256) v = xyzz_add(v, load_xyzz(&partials[(size_t)j * count + k]));
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = xyzz_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) store_xyzz(&window_sums[j], sh[0]);
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
    const u32* w = (const u32*)&a;
    uint4* p = out + i * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

template <class SF, class BF>
int msm_enqueue_t(const void* bases_dev, const void* scalars_dev, size_t n, size_t batch, size_t stride, int mont, hipStream_t s) {
    Ctx& c = ctx();
    MsmScratch& m = c.msm;
    const int cb = choose_window_bits(n);
    const int W = num_windows(cb);
    const u32 nbk = 1u << (cb - 1), nb1 = nbk + 1;
    // reduce geometry
    u32 tpw = nbk < 2048 ? nbk : 2048;  // threads per window
    if (tpw > nbk) tpw = nbk;
    const u32 slice = nbk / tpw;
    const u32 rblocks = (tpw + 255) / 256;

    TRH_TRY(m.digits.ensure((size_t)W * n * 4 + 16));
    TRH_TRY(m.sorted.ensure((size_t)W * n * 4 + 16));
    TRH_TRY(m.counts.ensure((size_t)W * nb1 * 4));
    TRH_TRY(m.starts.ensure((size_t)W * nb1 * 4));
    TRH_TRY(m.buckets.ensure((size_t)W * nbk * sizeof(XYZZ<BF>)));
    TRH_TRY(m.partials.ensure((size_t)W * rblocks * sizeof(XYZZ<BF>)));
    TRH_TRY(m.window_sums.ensure(batch * W * sizeof(XYZZ<BF>)));
    const size_t hs = batch * W * sizeof(XYZZ<BF>);
    if (hs > m.host_sums_cap) {
        if (m.host_sums) (void)hipHostFree(m.host_sums);
        TRH_HIP_TRY(hipHostMalloc(&m.host_sums, hs + 4096, hipHostMallocDefault));
        m.host_sums_cap = hs + 4096;
    }
    const bool timing = c.timing && batch == 1;
    if (timing && !m.ev[0]) for (int k = 0; k < 6; ++k) TRH_HIP_TRY(hipEventCreate(&m.ev[k]));

    for (size_t bi = 0; bi < batch; ++bi) {
        const uint4* sc = (const uint4*)((const char*)scalars_dev + bi * stride * 32);
        if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[0], s));
        if (n) {
            TRH_HIP_TRY(hipMemsetAsync(m.counts.p, 0, (size_t)W * nb1 * 4, s));
            const unsigned gb = (unsigned)((n + 255) / 256);
            hipLaunchKernelGGL((msm_recode_kernel<SF>), dim3(gb), dim3(256), 0, s, sc, n, mont, cb, W, m.digits.as<u32>(), m.counts.as<u32>(), nb1);
            if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[1], s));
            hipLaunchKernelGGL(msm_offsets_kernel, dim3(W), dim3(1024), 0, s, m.counts.as<u32>(), m.starts.as<u32>(), nb1);
            hipLaunchKernelGGL(msm_scatter_kernel, dim3(gb, W), dim3(256), 0, s, m.digits.as<u32>(), m.counts.as<u32>(), m.sorted.as<u32>(), n, nb1);
            if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[2], s));
            hipLaunchKernelGGL((msm_accumulate_kernel<BF>), dim3((nbk + 255) / 256, W), dim3(256), 0, s, (const uint4*)bases_dev, m.sorted.as<u32>(),
                               m.starts.as<u32>(), m.counts.as<u32>(), m.buckets.as<XYZZ<BF>>(), n, nbk);
            if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[3], s));
            hipLaunchKernelGGL((msm_reduce_kernel<BF>), dim3(rblocks, W), dim3(256), 0, s, m.buckets.as<XYZZ<BF>>(), m.partials.as<XYZZ<BF>>(), nbk, slice, tpw);
            hipLaunchKernelGGL((msm_window_sum_kernel<BF>), dim3(W), dim3(256), 0, s, m.partials.as<XYZZ<BF>>(), m.window_sums.as<XYZZ<BF>>() + bi * W, rblocks);
        } else {
            TRH_HIP_TRY(hipMemsetAsync(m.window_sums.as<XYZZ<BF>>() + bi * W, 0, W * sizeof(XYZZ<BF>), s));
            if (timing) for (int k = 1; k <= 3; ++k) TRH_HIP_TRY(hipEventRecord(m.ev[k], s));
        }
        if (timing) TRH_HIP_TRY(hipEventRecord(m.ev[4], s));
    }
    TRH_HIP_TRY(hipGetLastError());
    TRH_HIP_TRY(hipMemcpyAsync(m.host_sums, m.window_sums.p, hs, hipMemcpyDeviceToHost, s));
    m.pending_curve = BF::ID;
    m.pending_windows = W;
    m.pending_c = cb;
    m.pending_batch = batch;
    m.ev_valid = timing;
    return TRH_OK;
}

// host: Horner over windows, normalise
template <class BF>
void combine_windows_host(const XYZZ<BF>* ws, int W, int cb, u64* out_xyz) {
    XYZZ<BF> acc = xyzz_identity<BF>();
    for (int j = W - 1; j >= 0; --j) {
        for (int k = 0; k < cb; ++k) acc = xyzz_dbl(acc);
        acc = xyzz_add(acc, ws[j]);
    }
    Jacobian<BF> r = jac_from_affine(xyzz_to_affine(acc));
    memcpy(out_xyz, &r, 96);
}

template <class BF>
int msm_finish_t(hipStream_t s, u64* out_xyz, size_t batch) {
    Ctx& c = ctx();
    MsmScratch& m = c.msm;
    if (m.pending_curve != BF::ID || m.pending_batch != batch) { set_error("msm_finish: no matching MSM enqueued"); return TRH_EINVAL; }
    TRH_HIP_TRY(hipStreamSynchronize(s));
    const XYZZ<BF>* ws = (const XYZZ<BF>*)m.host_sums;
    for (size_t bi = 0; bi < batch; ++bi) combine_windows_host<BF>(ws + bi * m.pending_windows, m.pending_windows, m.pending_c, out_xyz + 12 * bi);
    if (m.ev_valid) {
        float t01, t12, t23, t34, tt;
        TRH_HIP_TRY(hipEventElapsedTime(&t01, m.ev[0], m.ev[1]));
        TRH_HIP_TRY(hipEventElapsedTime(&t12, m.ev[1], m.ev[2]));
        TRH_HIP_TRY(hipEventElapsedTime(&t23, m.ev[2], m.ev[3]));
        TRH_HIP_TRY(hipEventElapsedTime(&t34, m.ev[3], m.ev[4]));
        TRH_HIP_TRY(hipEventElapsedTime(&tt, m.ev[0], m.ev[4]));
        c.last.total_ms = tt; c.last.digits_ms = t01; c.last.sort_ms = t12; c.last.accumulate_ms = t23; c.last.reduce_ms = t34;
    }
    c.last.window_bits = m.pending_c;
    c.last.windows = m.pending_windows;
    m.pending_curve = -1;
    return TRH_OK;
}

template <class BF>
int point_sum_host_t(const u64* pts, size_t count, u64* out) {
    XYZZ<BF> acc = xyzz_identity<BF>();
    for (size_t i = 0; i < count; ++i) {
        Jacobian<BF> j;
        memcpy(&j, pts + 12 * i, 96);
        acc = xyzz_add(acc, xyzz_from_jacobian(j));
    }
    Jacobian<BF> r = jac_from_affine(xyzz_to_affine(acc));
    memcpy(out, &r, 96);
    return TRH_OK;
}

}  // namespace

int msm_enqueue(int curve, const void* bases_dev, const void* scalars_dev, size_t n, size_t batch, size_t stride, int mont, hipStream_t s) {
    // pallas: base Fp, scalar Fq; vesta: base Fq, scalar Fp
    if (curve == TRH_PALLAS) return msm_enqueue_t<FqParams, FpParams>(bases_dev, scalars_dev, n, batch, stride, mont, s);
    return msm_enqueue_t<FpParams, FqParams>(bases_dev, scalars_dev, n, batch, stride, mont, s);
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
    m.scalars.release(); m.digits.release(); m.sorted.release(); m.counts.release(); m.starts.release();
    m.buckets.release(); m.partials.release(); m.window_sums.release();
    if (m.host_sums) (void)hipHostFree(m.host_sums);
    m.host_sums = nullptr; m.host_sums_cap = 0;
    for (int k = 0; k < 6; ++k) if (m.ev[k]) { (void)hipEventDestroy(m.ev[k]); m.ev[k] = nullptr; }
}

}  // namespace trh
