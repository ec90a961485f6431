// The generator collapse of the IPA opening, r rounds at once, over the fixed-base table of g || w || u (round 6).
//
// halo2_proofs 0.2.0 `poly::commitment::prover::create_proof` folds its generators after every round:
//     G'[i] = G'[i] + u_j G'[i + half]        (parallel_generator_collapse; reference call site /root/reference/src/test_utils.rs:41-49)
// ipa.hip never materialises G': its rounds are MSMs over the ORIGINAL 2^k bases with the folds kept as weights, which makes every round
// cost a full 2^k MSM (0.62 ms at k = 18) however short p' has become.  After r rounds the folded generators are
//     G''[i] = sum over t < 2^r of s_t G[i + t m],     m = 2^(k - r),     s_t = product of u_j over the set bits (r - 1 - j) of t,
// i.e. m sums of 2^r points whose 2^r SCALARS ARE THE SAME for every i.  With the table T[j][x] = 2^(c j) G[x] each s_t splits into W
// chunks of c bits, each chunk into two signed sub-digits of <= 9 bits (8 + 8 for the c = 16 table of k = 18), and
//     G''[i] = sum over sub-windows s of 2^(shift_s) * sum over d of d * B[s][d][i],
//     B[s][d][i] = sum over the (t, j) whose sub-digit (j, s) is +-d of +-T[j][i + t m].
// The lists of (t, j, sign) per bucket (s, d) do not depend on i: the host builds them from the 2^r scalars (a counting sort of 2^r W 2
// entries), and the accumulation kernel needs NO SORT and NO GATHER -- lane i of a wave walks the wave's list and reads T[j][i + t m], 64
// consecutive 128-byte records per step.  2^k W 2 mixed additions in all (8.4 M at k = 18: what ONE unfolded round costs, twice), then
// 2 m running sums over 2^(w-1) buckets in slices, one inversion per generator.  The opening's remaining k - r rounds run over the m + 2
// points G'' || w || u (ipa.hip).
#include <string.h>

#include <vector>

#include "ctx.h"
#include "hostcombine.h"

namespace trh {
namespace {

constexpr u32 FOLD_SIGN = 0x80000000u;

template <class BF>
__device__ __forceinline__ void fold_store_raw(XYZZzMem* dst, const XYZZz<BF>& v) {
    uint4* p = (uint4*)dst;
    const u32* w = (const u32*)&v;
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
template <class BF>
__device__ __forceinline__ XYZZz<BF> fold_load_raw(const XYZZzMem* src) {
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

// plan = offsets[nbk + 1] then entries: level j << 16 | t | sign << 31 (host: fold_plan)
// grid (m / 256, nbk): every lane of a workgroup walks the SAME list -- the control flow is uniform and the loads are contiguous over i
template <class BF>
__global__ void __launch_bounds__(256) ipa_fold_accumulate_kernel(const uint4* __restrict__ table, size_t row /* points per table level */, u32 log_m,
                                                                  const u32* __restrict__ plan, u32 nbk, XYZZzMem* __restrict__ out) {
    const u32 b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 lo = plan[b], hi = plan[b + 1];
    const u32* ent = plan + nbk + 1;
    XYZZz<BF> acc = xyzzz_identity<BF>();
    bool fresh = true;
    struct Slot { u32 e; uint4 a, b, c, d, t; };
    auto issue = [&](Slot& sl, u32 entry) {
        sl.e = entry;
        const uint4* bp = table + (((size_t)((entry >> 16) & 0x7FFFu)) * row + ((size_t)(entry & 0xFFFFu) << log_m) + i) * (ZREC / 16);
        const uint4* yp = bp + 2 + ((entry >> 31) << 1);  // y, or -y for a negative digit
        sl.a = bp[0]; sl.b = bp[1]; sl.c = yp[0]; sl.d = yp[1]; sl.t = bp[6];
    };
    auto unpack = [&](const Slot& sl, AffineZ<BF>& p) {
        p.x.l[0] = (i32)sl.a.x; p.x.l[1] = (i32)sl.a.y; p.x.l[2] = (i32)sl.a.z; p.x.l[3] = (i32)sl.a.w;
        p.x.l[4] = (i32)sl.b.x; p.x.l[5] = (i32)sl.b.y; p.x.l[6] = (i32)sl.b.z; p.x.l[7] = (i32)sl.b.w; p.x.l[8] = (i32)sl.t.x;
        p.y.l[0] = (i32)sl.c.x; p.y.l[1] = (i32)sl.c.y; p.y.l[2] = (i32)sl.c.z; p.y.l[3] = (i32)sl.c.w;
        p.y.l[4] = (i32)sl.d.x; p.y.l[5] = (i32)sl.d.y; p.y.l[6] = (i32)sl.d.z; p.y.l[7] = (i32)sl.d.w; p.y.l[8] = (i32)((sl.e >> 31) ? sl.t.z : sl.t.y);
    };
    if (hi > lo) {
        const u32 last_pos = hi - 1;
        auto clamp = [&](u32 q) { return q < last_pos ? q : last_pos; };
        u32 pos = lo;
        Slot s0, s1;
        issue(s0, ent[pos]);
        issue(s1, ent[clamp(pos + 1)]);
        auto step = [&](Slot& sl) {
            const u32 ce = sl.e;
            AffineZ<BF> cur;
            unpack(sl, cur);
            const bool p_identity = ((cur.y.l[0] | cur.y.l[1] | cur.y.l[2]) | (cur.y.l[3] | cur.y.l[4] | cur.y.l[5]) | (cur.y.l[6] | cur.y.l[7] | cur.y.l[8])) == 0;
            issue(sl, ent[clamp(pos + 2)]);  // two records in flight, as in msm_accumulate_seg_kernel
            if (!p_identity) {
                if (fresh) {
                    acc.x = cur.x; acc.y = cur.y; acc.zz = fy_one<BF>(); acc.zzz = fy_one<BF>();
                    fresh = false;
                } else {
                    Fy<BF> R;
                    const bool same_x = xyzzz_madd_main(acc, cur, R);
                    if (__any(same_x)) {
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
            }
            ++pos;
        };
        while (pos + 1 < hi) {
            step(s0);
            step(s1);
        }
        if (pos < hi) step(s0);
    }
    if (fresh) acc = xyzzz_identity<BF>();
    fold_store_raw(out + ((size_t)b << log_m) + i, acc);
}

// 32 lanes per generator: lane = (sub-window s, slice q < 16).  Slice: running sums over its NB / 16 buckets, + (first id - 1) * (sum of the
// slice); a shuffle tree over the sixteen slices; the high sub-window's sum doubled w0 times onto the low one.  The sum leaves as a raw point:
// the inversion has a kernel of its own (in here every wave would walk the whole exponent for the two lanes that hold a result).
// (First version: 16 lanes, slices of 16 buckets, the inversion inline -- 0.73 ms at k = 18, one wave per SIMD waiting on itself.)
template <class BF>
__global__ void __launch_bounds__(256) ipa_fold_reduce_kernel(const XYZZzMem* __restrict__ buckets, u32 log_m, u32 w0, u32 w1, XYZZzMem* __restrict__ sums) {
    const u32 lane32 = threadIdx.x & 31u;
    const size_t i = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const u32 s = lane32 >> 4, q = lane32 & 15u;
    const u32 nb = 1u << ((s ? w1 : w0) - 1);
    const u32 sl = nb >> 4;  // buckets per slice (w >= 5)
    const u32 first = (s ? (1u << (w0 - 1)) : 0u) + q * sl;  // row of the slice's first bucket
    XYZZz<BF> run = xyzzz_identity<BF>(), acc = xyzzz_identity<BF>();
    for (int k = (int)sl - 1; k >= 0; --k) {
        const XYZZz<BF> v = fold_load_raw<BF>(buckets + ((size_t)(first + k) << log_m) + i);
        run = xyzzz_add(run, v);
        acc = xyzzz_add(acc, run);
    }
    const u32 off = q * sl;  // the slice's first bucket has digit off + 1
    if (off) {
        XYZZz<BF> sc = xyzzz_identity<BF>();
        for (int bit = 31 - __clz(off); bit >= 0; --bit) {
            sc = xyzzz_dbl(sc);
            if ((off >> bit) & 1u) sc = xyzzz_add(sc, run);
        }
        acc = xyzzz_add(acc, sc);
    }
    for (int d = 8; d > 0; d >>= 1) {
        XYZZz<BF> o;
#pragma unroll
        for (int l = 0; l < NLIMBS; ++l) {
            o.x.l[l] = __shfl_down(acc.x.l[l], d, 16); o.y.l[l] = __shfl_down(acc.y.l[l], d, 16);
            o.zz.l[l] = __shfl_down(acc.zz.l[l], d, 16); o.zzz.l[l] = __shfl_down(acc.zzz.l[l], d, 16);
        }
        if ((int)q < d) acc = xyzzz_add(acc, o);
    }
    if (lane32 == 16) for (u32 t = 0; t < w0; ++t) acc = xyzzz_dbl(acc);
    {
        XYZZz<BF> o;
#pragma unroll
        for (int l = 0; l < NLIMBS; ++l) {
            o.x.l[l] = __shfl_down(acc.x.l[l], 16, 32); o.y.l[l] = __shfl_down(acc.y.l[l], 16, 32);
            o.zz.l[l] = __shfl_down(acc.zz.l[l], 16, 32); o.zzz.l[l] = __shfl_down(acc.zzz.l[l], 16, 32);
        }
        if (lane32 != 0) return;
        acc = xyzzz_add(acc, o);
    }
    fold_store_raw(sums + i, acc);
}

// a thread per generator: affine in the lazy domain -- 1 / zzz by Fermat with this domain's products (255 squarings + ~65 products: the moduli
// are 2^254 + a 126-bit number), 1 / zz = zzz^-2 zz^2 -- and both base formats (64-byte affine, msm.hip's 128-byte record)
template <class BF>
__global__ void __launch_bounds__(64) ipa_fold_affine_kernel(const XYZZzMem* __restrict__ sums, size_t m, uint4* __restrict__ out_xy, uint4* __restrict__ out_z) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= m) return;
    const XYZZz<BF> acc = fold_load_raw<BF>(sums + i);
    Affine<BF> a;
    if (xyzzz_is_identity(acc)) { a.x = fe_zero<BF>(); a.y = fe_zero<BF>(); }  // (never, for independent generators)
    else {
        Fy<BF> inv = acc.zzz;  // exponent m - 2: top bit, then bits 253 .. 0
        for (int bit = 253; bit >= 0; --bit) {
            inv = fy_sqr(inv);
            const int wi = bit >> 5;  // m - 2: the low word of m is 1, so word 0 becomes 0xffffffff and word 1 lends one (as fe_inv)
            const u32 word = wi == 0 ? 0xffffffffu : wi == 1 ? BF::MOD[1] - 1u : BF::MOD[wi];
            if ((word >> (bit & 31)) & 1u) inv = fy_mul(inv, acc.zzz);
        }
        const Fy<BF> zz_inv = fy_mul(fy_sqr(inv), fy_sqr(acc.zz));
        a.x = fy_to_fe(fy_mul(acc.x, zz_inv));
        a.y = fy_to_fe(fy_mul(acc.y, inv));
    }
    u32 w[8];
    uint4* p = out_xy + i * 4;
    fe_store(a.x, w); p[0] = make_uint4(w[0], w[1], w[2], w[3]); p[1] = make_uint4(w[4], w[5], w[6], w[7]);
    fe_store(a.y, w); p[2] = make_uint4(w[0], w[1], w[2], w[3]); p[3] = make_uint4(w[4], w[5], w[6], w[7]);
    const Fy<BF> x = fy_from_fe(a.x), y = fy_from_fe(a.y);
    uint4* z = out_z + i * (ZREC / 16);  // the record of msm.hip's store_zrec
    z[0] = make_uint4((u32)x.l[0], (u32)x.l[1], (u32)x.l[2], (u32)x.l[3]); z[1] = make_uint4((u32)x.l[4], (u32)x.l[5], (u32)x.l[6], (u32)x.l[7]);
    z[2] = make_uint4((u32)y.l[0], (u32)y.l[1], (u32)y.l[2], (u32)y.l[3]); z[3] = make_uint4((u32)y.l[4], (u32)y.l[5], (u32)y.l[6], (u32)y.l[7]);
    z[4] = make_uint4((u32)-y.l[0], (u32)-y.l[1], (u32)-y.l[2], (u32)-y.l[3]); z[5] = make_uint4((u32)-y.l[4], (u32)-y.l[5], (u32)-y.l[6], (u32)-y.l[7]);
    z[6] = make_uint4((u32)x.l[8], (u32)y.l[8], (u32)-y.l[8], 0u);
    z[7] = make_uint4(0u, 0u, 0u, 0u);
}

// the bucket lists of the 2^r shared scalars (canonical 4 x 64-bit words each): offsets[nbk + 1], then the entries bucket by bucket
int fold_plan(const std::vector<hostcombine::H>& sc, int c, int W, u32 w0, u32 w1, std::vector<u32>& plan, u32& nbk) {
    const u32 nb0 = 1u << (w0 - 1), nb1 = 1u << (w1 - 1);
    nbk = nb0 + nb1;
    std::vector<u32> bucket, entry;
    bucket.reserve(sc.size() * W * 2); entry.reserve(sc.size() * W * 2);
    std::vector<u32> count(nbk + 1, 0);
    auto bits = [](const hostcombine::H& v, u32 o, u32 w) -> u32 {
        if (o >= 256) return 0;
        uint64_t x = v.l[o >> 6] >> (o & 63);
        if ((o & 63) + w > 64 && (o >> 6) + 1 < 4) x |= v.l[(o >> 6) + 1] << (64 - (o & 63));
        return (u32)(x & ((1ull << w) - 1));
    };
    for (size_t t = 0; t < sc.size(); ++t) {
        u32 carry = 0;
        for (int j = 0; j < W; ++j)
            for (u32 s = 0; s < 2; ++s) {
                const u32 w = s ? w1 : w0;
                int d = (int)(bits(sc[t], (u32)(c * j) + (s ? w0 : 0), w) + carry);
                if (d > (1 << (w - 1))) { d -= 1 << w; carry = 1; } else carry = 0;
                if (!d) continue;
                const u32 mag = (u32)(d < 0 ? -d : d);
                const u32 b = (s ? nb0 : 0) + mag - 1;
                bucket.push_back(b);
                entry.push_back(((u32)j << 16) | (u32)t | (d < 0 ? FOLD_SIGN : 0u));
                ++count[b];
            }
        if (carry) { set_error("ipa fold: a fold scalar does not fit the table's %d windows of %d bits", W, c); return TRH_EINVAL; }
    }
    plan.assign(nbk + 1 + entry.size(), 0);
    u32 run = 0;
    for (u32 b = 0; b < nbk; ++b) { plan[b] = run; run += count[b]; }
    plan[nbk] = run;
    std::vector<u32> cur(plan.begin(), plan.begin() + nbk);
    for (size_t e = 0; e < entry.size(); ++e) plan[nbk + 1 + cur[bucket[e]]++] = entry[e];
    return TRH_OK;
}

// buckets (+ the m sums between the reduction and the inversion), the bucket lists and their pinned source
int fold_buffers(Ctx& c, const MsmFixedBase& fb, uint32_t k, uint32_t r, size_t plan_words, hipStream_t s) {
    const size_t m = (size_t)1 << (k - r);
    const u32 nbk = (1u << ((fb.c + 1) / 2 - 1)) + (1u << (fb.c / 2 - 1));
    TRH_TRY(c.ipa[9].ensure((size_t)(nbk + 1) * m * sizeof(XYZZzMem)));
    TRH_TRY(c.ipa[10].ensure(plan_words * 4));
    if (c.pinned_fold_cap < plan_words * 4) {
        if (c.pinned_fold) { TRH_HIP_TRY(hipStreamSynchronize(s)); (void)hipHostFree(c.pinned_fold); c.pinned_fold = nullptr; c.pinned_fold_cap = 0; }
        const size_t want = plan_words * 4 + 4096;
        TRH_HIP_TRY(hipHostMalloc(&c.pinned_fold, want, hipHostMallocDefault));
        c.pinned_fold_cap = want;
    }
    return TRH_OK;
}

template <class SF, class BF>
int ipa_fold_t(const MsmFixedBase& fb, size_t row, uint32_t k, uint32_t r, const u64* u_mont, void* out_xy, void* out_z, hipStream_t s) {
    using hostcombine::H;
    Ctx& c = ctx();
    const u32 log_m = k - r;
    const size_t m = (size_t)1 << log_m;
    const u32 w0 = (u32)(fb.c + 1) / 2, w1 = (u32)fb.c / 2;
    // s_t (Montgomery) by doubling the set of indices: bit (r - 1 - j) of t <-> u_j
    std::vector<H> sc((size_t)1 << r);
    sc[0] = hostcombine::consts<SF>().one;
    for (uint32_t j = 0; j < r; ++j) {
        H u;
        memcpy(&u, u_mont + 4 * j, 32);
        const size_t bit = (size_t)1 << (r - 1 - j);
        // the indices built so far are the multiples of 2 * bit: each gets its sibling with `bit` set
        for (size_t t = 0; t < sc.size(); t += 2 * bit) sc[t + bit] = hostcombine::mul<SF>(sc[t], u);
    }
    const H one_plain = {{1, 0, 0, 0}};
    for (H& v : sc) v = hostcombine::mul<SF>(v, one_plain);  // out of the Montgomery form: canonical words
    std::vector<u32> plan;
    u32 nbk = 0;
    TRH_TRY(fold_plan(sc, fb.c, fb.W, w0, w1, plan, nbk));
    DevBuf& buckets = c.ipa[9];
    DevBuf& dplan = c.ipa[10];
    TRH_TRY(fold_buffers(c, fb, k, r, plan.size(), s));
    XYZZzMem* const sums = (XYZZzMem*)buckets.p + (size_t)nbk * m;
    // (the previous opening's copy out of this buffer completed before that opening returned: it ends with a stream synchronisation)
    memcpy(c.pinned_fold, plan.data(), plan.size() * 4);
    TRH_HIP_TRY(hipMemcpyAsync(dplan.p, c.pinned_fold, plan.size() * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((ipa_fold_accumulate_kernel<BF>), dim3((unsigned)(m / 256), nbk), dim3(256), 0, s, (const uint4*)fb.table, row, log_m, (const u32*)dplan.p, nbk,
                       (XYZZzMem*)buckets.p);
    hipLaunchKernelGGL((ipa_fold_reduce_kernel<BF>), dim3((unsigned)(m / 8)), dim3(256), 0, s, (const XYZZzMem*)buckets.p, log_m, w0, w1, sums);
    hipLaunchKernelGGL((ipa_fold_affine_kernel<BF>), dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, (const XYZZzMem*)sums, m, (uint4*)out_xy, (uint4*)out_z);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

}  // namespace

bool ipa_fold_supported(const MsmFixedBase& fb, uint32_t k, uint32_t r) {
    // two sub-digits of 5 .. 9 bits per table window (sixteen slices of >= 1 bucket); t and the table level share an entry word; whole workgroups of generators
    return fb.table && fb.c >= 10 && fb.c <= 18 && fb.W < 0x8000 && r >= 2 && r <= 10 && k >= r + 8;
}

// setup-time sizing of what ipa_fold_generators allocates (trh_bases_reserve over an opening's base set)
int ipa_fold_reserve(const MsmFixedBase& fb, uint32_t k, uint32_t r) {
    if (!ipa_fold_supported(fb, k, r)) return TRH_OK;
    const u32 nbk = (1u << ((fb.c + 1) / 2 - 1)) + (1u << (fb.c / 2 - 1));
    return fold_buffers(ctx(), fb, k, r, (size_t)nbk + 1 + ((size_t)2 << r) * fb.W, nullptr);
}

// G''[i] (i < 2^(k - r)) from the table of a base set with `row` points per level, after the r challenges u_mont (Montgomery words, round order):
// out_xy = m affine points (64-byte records), out_z = the same in the accumulation's 128-byte form.  Enqueued on s; the caller holds the context.
int ipa_fold_generators(int curve, const MsmFixedBase& fb, size_t row, uint32_t k, uint32_t r, const u64* u_mont, void* out_xy, void* out_z, hipStream_t s) {
    if (!ipa_fold_supported(fb, k, r)) { set_error("ipa fold: unsupported shape (table window %d, k = %u, r = %u)", fb.c, k, r); return TRH_EINVAL; }
    if (curve == TRH_PALLAS) return ipa_fold_t<FqParams, FpParams>(fb, row, k, r, u_mont, out_xy, out_z, s);
    return ipa_fold_t<FpParams, FqParams>(fb, row, k, r, u_mont, out_xy, out_z, s);
}

}  // namespace trh
