// The permuted columns of halo2_proofs 0.2.0's lookup argument (plonk/lookup/prover.rs `permute_expression_pair`, reached
// from create_proof -- /root/reference/src/test_utils.rs:41-49; the reference's circuit has 31 lookups,
// src/circuits/even_bits.rs:158-170, aux/out_table.rs:33-74, shift.rs:142-165, tables/prog.rs:170-192):
//
//   A' = the first `usable_rows` input values sorted (Ord of the field = order of the canonical integers)
//   S'[row] = A'[row] where A'[row] is the first of its run; one instance of that value leaves the table multiset;
//   the other rows receive the left-over table values in ascending order, the LAST repeated row first
//   (the Rust code pops the rows of a Vec and walks a BTreeMap);  an input value missing from the table is an error.
//
// A sort of 256-bit keys, not a hot kernel of this path: the 64-bit limb sorts are rocPRIM's radix sort (four stable
// least-significant-limb-first passes over (limb, index) pairs), everything else (canonical form, run flags, membership
// by binary search, compaction by prefix sums, the final placement) is a handful of small kernels here.
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "ctx.h"

namespace trh {
namespace {

template <class F>
__device__ __forceinline__ Fe<F> ldf(const uint4* p) {
    uint4 a = p[0], b = p[1];
    return fe_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}

// canonical limbs of a[i] (Montgomery -> integer), one u64 plane per limb; perm = identity
template <class F>
__global__ void __launch_bounds__(256) canon_planes_kernel(const uint4* __restrict__ a, size_t n, u64* __restrict__ planes /* 4 x n */, u32* __restrict__ perm) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    fe_store(fe_from_mont(ldf<F>(a + 2 * i)), w);
    for (int k = 0; k < 4; ++k) planes[(size_t)k * n + i] = (u64)w[2 * k] | ((u64)w[2 * k + 1] << 32);
    perm[i] = (u32)i;
}
__global__ void __launch_bounds__(256) gather_u64_kernel(const u64* __restrict__ src, const u32* __restrict__ perm, u64* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
__global__ void __launch_bounds__(256) gather_elems_kernel(const uint4* __restrict__ src, const u32* __restrict__ perm, uint4* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* p = src + 2 * (size_t)perm[i];
    dst[2 * i] = p[0]; dst[2 * i + 1] = p[1];
}
// sorted canonical keys as 4 planes gathered through perm: cmp(i, j) on the planes
__device__ __forceinline__ int cmp_keys(const u64* __restrict__ pa, size_t na, const u32* __restrict__ perma, size_t i, const u64* __restrict__ pb, size_t nb, const u32* __restrict__ permb, size_t j) {
    const u32 ia = perma[i], ib = permb[j];
    for (int k = 3; k >= 0; --k) {
        const u64 x = pa[(size_t)k * na + ia], y = pb[(size_t)k * nb + ib];
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
// first[i] = 1 when sorted element i starts a run of equal values
__global__ void __launch_bounds__(256) run_flags_kernel(const u64* __restrict__ planes, size_t n, const u32* __restrict__ perm, u32* __restrict__ first) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    first[i] = (i == 0 || cmp_keys(planes, n, perm, i, planes, n, perm, i - 1) != 0) ? 1u : 0u;
}
// every run start of the sorted input removes the first table instance of its value: removed[pos] = 1; missing -> *err = 1
__global__ void __launch_bounds__(256) remove_from_table_kernel(const u64* __restrict__ pa, const u32* __restrict__ perma, const u32* __restrict__ first, size_t n,
                                                                const u64* __restrict__ ps, const u32* __restrict__ perms, u32* __restrict__ removed, u32* __restrict__ err) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !first[i]) return;
    size_t lo = 0, hi = n;  // lower bound of the value in the sorted table
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (cmp_keys(ps, n, perms, mid, pa, n, perma, i) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo < n && cmp_keys(ps, n, perms, lo, pa, n, perma, i) == 0) removed[lo] = 1u;  // distinct run starts hit distinct positions
    else *err = 1u;
}
__global__ void __launch_bounds__(256) invert_flags_kernel(const u32* __restrict__ in, u32* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] ? 0u : 1u;
}
// rows[] of the flagged positions in ascending order (pos = exclusive prefix sum of the flags)
__global__ void __launch_bounds__(256) compact_kernel(const u32* __restrict__ flags, const u32* __restrict__ pos, u32* __restrict__ rows, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) rows[pos[i]] = (u32)i;
}
// S'[row] = A'[row] at run starts; left-over table element k (ascending) goes to the (count - 1 - k)-th repeated row
__global__ void __launch_bounds__(256) place_table_kernel(const uint4* __restrict__ a_sorted, const u32* __restrict__ first, const uint4* __restrict__ table, const u32* __restrict__ perms,
                                                          const u32* __restrict__ left_rows /* positions in the sorted table */, const u32* __restrict__ rep_rows, u32 n_rep,
                                                          uint4* __restrict__ out_table, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (first[i]) { out_table[2 * i] = a_sorted[2 * i]; out_table[2 * i + 1] = a_sorted[2 * i + 1]; }
    if (i < n_rep) {
        const uint4* src = table + 2 * (size_t)perms[left_rows[i]];
        const size_t dst = rep_rows[n_rep - 1 - i];
        out_table[2 * dst] = src[0]; out_table[2 * dst + 1] = src[1];
    }
}

// varies[k] = 1 when limb plane k is not constant over the column
__global__ void __launch_bounds__(256) plane_varies_kernel(const u64* __restrict__ planes, size_t n, u32* __restrict__ varies) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 4; ++k)
        if (planes[(size_t)k * n + i] != planes[(size_t)k * n] && !varies[k]) varies[k] = 1u;  // benign race: every writer stores 1
}
// after a sort by ONE limb: *bad = 1 when two neighbours agree in that limb but are different values (the order inside the tie is then unknown)
__global__ void __launch_bounds__(256) tie_check_kernel(const u64* __restrict__ planes, size_t n, const u32* __restrict__ perm, int primary, u32* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 || i >= n) return;
    const u32 a = perm[i - 1], b = perm[i];
    if (planes[(size_t)primary * n + a] != planes[(size_t)primary * n + b]) return;
    for (int k = 0; k < 4; ++k)
        if (planes[(size_t)k * n + a] != planes[(size_t)k * n + b]) { *bad = 1u; return; }
}

struct Scratch {
    DevBuf planes_a, planes_s, keys_in, keys_out, perm_a, perm_s, perm_tmp, first, removed, flags, pos, rows_rep, rows_left, tmp, err;
    u32* host = nullptr;  // pinned landing area of the few words the host reads back per lookup (pageable targets cost a staging copy each)
};
Scratch& scratch() {  // per context (device)
    Ctx& c = ctx();
    if (!c.lookup_scratch) c.lookup_scratch = new Scratch();
    return *(Scratch*)c.lookup_scratch;
}
int host_words(Scratch& sc) {
    if (!sc.host) TRH_HIP_TRY(hipHostMalloc((void**)&sc.host, 256, hipHostMallocDefault));
    return TRH_OK;
}
// the three words the placement needs: exclusive-scan tail and last flag of the repeated rows, the missing-value flag
__global__ void collect_tail_kernel(const u32* __restrict__ pos, const u32* __restrict__ flags, size_t n, const u32* __restrict__ err, u32* __restrict__ out) {
    out[0] = pos[n - 1]; out[1] = flags[n - 1]; out[2] = err[0];
}

// sort of the permutations of BOTH columns by their 256-bit keys.  Field elements are either small (range tables: only the low
// limb varies) or spread over the whole field (compressed expressions: two different values practically never share their top
// limb), so ONE radix sort by the most significant limb that varies is almost always the complete order -- checked, with the
// stable least-significant-limb-first passes as the fallback.  The two columns go through the phases together so that the
// host reads the flags of both with one synchronisation per phase.
int sort_perm_fallback(const u64* planes, size_t n, u32* perm, const u32* varies, size_t tmp_bytes, hipStream_t s) {
    Scratch& sc = scratch();
    const unsigned gb = (unsigned)((n + 255) / 256);
    u32* cur = perm;  // still the identity order
    u32* nxt = sc.perm_tmp.as<u32>();
    for (int k = 0; k < 4; ++k) {
        if (!varies[k]) continue;
        hipLaunchKernelGGL(gather_u64_kernel, dim3(gb), dim3(256), 0, s, planes + (size_t)k * n, cur, sc.keys_in.as<u64>(), n);
        TRH_HIP_TRY(rocprim::radix_sort_pairs(sc.tmp.p, tmp_bytes, sc.keys_in.as<u64>(), sc.keys_out.as<u64>(), cur, nxt, n, 0, 64, s));
        u32* t = cur; cur = nxt; nxt = t;
    }
    if (cur != perm) TRH_HIP_TRY(hipMemcpyAsync(perm, cur, n * 4, hipMemcpyDeviceToDevice, s));
    return TRH_OK;
}

int sort_perm_pair(const u64* planes_a, u32* perm_a, const u64* planes_s, u32* perm_s, size_t n, hipStream_t s) {
    Scratch& sc = scratch();
    TRH_TRY(sc.keys_in.ensure(n * 8)); TRH_TRY(sc.keys_out.ensure(n * 8)); TRH_TRY(sc.perm_tmp.ensure(2 * n * 4)); TRH_TRY(sc.err.ensure(64));
    size_t tmp_bytes = 0;
    TRH_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, sc.keys_in.as<u64>(), sc.keys_out.as<u64>(), perm_a, sc.perm_tmp.as<u32>(), n, 0, 64, s));
    TRH_TRY(sc.tmp.ensure(tmp_bytes + 256));
    const unsigned gb = (unsigned)((n + 255) / 256);
    const u64* planes[2] = {planes_a, planes_s};
    u32* perms[2] = {perm_a, perm_s};
    u32* flags = sc.err.as<u32>() + 1;  // [0] is the caller's error word; per column: varies[4], bad
    TRH_HIP_TRY(hipMemsetAsync(flags, 0, 40, s));
    for (int c = 0; c < 2; ++c) hipLaunchKernelGGL(plane_varies_kernel, dim3(gb), dim3(256), 0, s, planes[c], n, flags + 5 * c);
    TRH_TRY(host_words(sc));
    u32 h[10];
    TRH_HIP_TRY(hipMemcpyAsync(sc.host, flags, 40, hipMemcpyDeviceToHost, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));
    memcpy(h, sc.host, 40);
    int primary[2];
    for (int c = 0; c < 2; ++c) {
        primary[c] = -1;
        for (int k = 3; k >= 0; --k) if (h[5 * c + k]) { primary[c] = k; break; }
        if (primary[c] < 0) continue;  // a constant column: any order
        u32* out = sc.perm_tmp.as<u32>() + (size_t)c * n;
        hipLaunchKernelGGL(gather_u64_kernel, dim3(gb), dim3(256), 0, s, planes[c] + (size_t)primary[c] * n, perms[c], sc.keys_in.as<u64>(), n);
        TRH_HIP_TRY(rocprim::radix_sort_pairs(sc.tmp.p, tmp_bytes, sc.keys_in.as<u64>(), sc.keys_out.as<u64>(), perms[c], out, n, 0, 64, s));
        hipLaunchKernelGGL(tie_check_kernel, dim3(gb), dim3(256), 0, s, planes[c], n, out, primary[c], flags + 5 * c + 4);
    }
    TRH_HIP_TRY(hipMemcpyAsync(sc.host, flags + 4, 24, hipMemcpyDeviceToHost, s));  // bad[0] ... bad[1]: words 4 and 9
    TRH_HIP_TRY(hipStreamSynchronize(s));
    const u32 bad[2] = {sc.host[0], sc.host[5]};
    for (int c = 0; c < 2; ++c) {
        if (primary[c] < 0) continue;
        if (!bad[c]) TRH_HIP_TRY(hipMemcpyAsync(perms[c], sc.perm_tmp.as<u32>() + (size_t)c * n, n * 4, hipMemcpyDeviceToDevice, s));
    }
    for (int c = 0; c < 2; ++c)  // after the copies: the fallback reuses perm_tmp
        if (primary[c] >= 0 && bad[c]) TRH_TRY(sort_perm_fallback(planes[c], n, perms[c], h + 5 * c, tmp_bytes, s));
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

int exclusive_scan_u32(const u32* in, u32* out, size_t n, hipStream_t s) {
    Scratch& sc = scratch();
    size_t tmp_bytes = 0;
    TRH_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, 0u, n, rocprim::plus<u32>(), s));
    TRH_TRY(sc.tmp.ensure(tmp_bytes + 256));
    TRH_HIP_TRY(rocprim::exclusive_scan(sc.tmp.p, tmp_bytes, in, out, 0u, n, rocprim::plus<u32>(), s));
    return TRH_OK;
}

template <class F>
int lookup_permute_t(const void* input, const void* table, size_t n, void* out_input, void* out_table, hipStream_t s) {
    Scratch& sc = scratch();
    TRH_TRY(sc.planes_a.ensure(n * 32)); TRH_TRY(sc.planes_s.ensure(n * 32)); TRH_TRY(sc.perm_a.ensure(n * 4)); TRH_TRY(sc.perm_s.ensure(n * 4));
    TRH_TRY(sc.first.ensure(n * 4)); TRH_TRY(sc.removed.ensure(n * 4)); TRH_TRY(sc.flags.ensure(n * 4)); TRH_TRY(sc.pos.ensure(n * 4 + 4));
    TRH_TRY(sc.rows_rep.ensure(n * 4)); TRH_TRY(sc.rows_left.ensure(n * 4)); TRH_TRY(sc.err.ensure(64));
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL((canon_planes_kernel<F>), dim3(gb), dim3(256), 0, s, (const uint4*)input, n, sc.planes_a.as<u64>(), sc.perm_a.as<u32>());
    hipLaunchKernelGGL((canon_planes_kernel<F>), dim3(gb), dim3(256), 0, s, (const uint4*)table, n, sc.planes_s.as<u64>(), sc.perm_s.as<u32>());
    TRH_TRY(sort_perm_pair(sc.planes_a.as<u64>(), sc.perm_a.as<u32>(), sc.planes_s.as<u64>(), sc.perm_s.as<u32>(), n, s));
    hipLaunchKernelGGL(gather_elems_kernel, dim3(gb), dim3(256), 0, s, (const uint4*)input, sc.perm_a.as<u32>(), (uint4*)out_input, n);
    hipLaunchKernelGGL(run_flags_kernel, dim3(gb), dim3(256), 0, s, sc.planes_a.as<u64>(), n, sc.perm_a.as<u32>(), sc.first.as<u32>());
    TRH_HIP_TRY(hipMemsetAsync(sc.removed.p, 0, n * 4, s));
    TRH_HIP_TRY(hipMemsetAsync(sc.err.p, 0, 4, s));
    hipLaunchKernelGGL(remove_from_table_kernel, dim3(gb), dim3(256), 0, s, sc.planes_a.as<u64>(), sc.perm_a.as<u32>(), sc.first.as<u32>(), n, sc.planes_s.as<u64>(), sc.perm_s.as<u32>(),
                       sc.removed.as<u32>(), sc.err.as<u32>());
    // repeated input rows (ascending) and left-over table positions (ascending)
    hipLaunchKernelGGL(invert_flags_kernel, dim3(gb), dim3(256), 0, s, sc.first.as<u32>(), sc.flags.as<u32>(), n);
    TRH_TRY(exclusive_scan_u32(sc.flags.as<u32>(), sc.pos.as<u32>(), n, s));
    hipLaunchKernelGGL(compact_kernel, dim3(gb), dim3(256), 0, s, sc.flags.as<u32>(), sc.pos.as<u32>(), sc.rows_rep.as<u32>(), n);
    TRH_TRY(host_words(sc));
    u32* rec = sc.err.as<u32>() + 12;  // device record behind the error word and the sort flags
    hipLaunchKernelGGL(collect_tail_kernel, dim3(1), dim3(1), 0, s, sc.pos.as<u32>(), sc.flags.as<u32>(), n, sc.err.as<u32>(), rec);
    hipLaunchKernelGGL(invert_flags_kernel, dim3(gb), dim3(256), 0, s, sc.removed.as<u32>(), sc.flags.as<u32>(), n);
    TRH_TRY(exclusive_scan_u32(sc.flags.as<u32>(), sc.pos.as<u32>(), n, s));
    hipLaunchKernelGGL(compact_kernel, dim3(gb), dim3(256), 0, s, sc.flags.as<u32>(), sc.pos.as<u32>(), sc.rows_left.as<u32>(), n);
    TRH_HIP_TRY(hipMemcpyAsync(sc.host, rec, 12, hipMemcpyDeviceToHost, s));
    TRH_HIP_TRY(hipStreamSynchronize(s));
    if (sc.host[2]) { set_error("lookup_permute: an input value does not occur in the table (halo2: Error::ConstraintSystemFailure)"); return TRH_EINVAL; }
    const u32 n_rep = sc.host[0] + sc.host[1];  // repeated rows == left-over table elements (both n - #distinct inputs)
    hipLaunchKernelGGL(place_table_kernel, dim3(gb), dim3(256), 0, s, (const uint4*)out_input, sc.first.as<u32>(), (const uint4*)table, sc.perm_s.as<u32>(), sc.rows_left.as<u32>(),
                       sc.rows_rep.as<u32>(), n_rep, (uint4*)out_table, n);
    TRH_HIP_TRY(hipGetLastError());
    return TRH_OK;
}

}  // namespace

void lookup_release() {
    Ctx& c = ctx();
    if (!c.lookup_scratch) return;
    Scratch& sc = *(Scratch*)c.lookup_scratch;
    for (DevBuf* b : {&sc.planes_a, &sc.planes_s, &sc.keys_in, &sc.keys_out, &sc.perm_a, &sc.perm_s, &sc.perm_tmp, &sc.first, &sc.removed, &sc.flags, &sc.pos, &sc.rows_rep, &sc.rows_left, &sc.tmp,
                      &sc.err})
        b->release();
    if (sc.host) { (void)hipHostFree(sc.host); sc.host = nullptr; }
    delete &sc;
    c.lookup_scratch = nullptr;
}

}  // namespace trh

using namespace trh;

extern "C" int trh_lookup_permute_dev(int field, const void* input_dev, const void* table_dev, size_t usable_rows, void* out_input_dev, void* out_table_dev, void* stream) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (usable_rows && (!input_dev || !table_dev || !out_input_dev || !out_table_dev)) { set_error("lookup_permute: null pointer"); return TRH_EINVAL; }
    if (out_input_dev == input_dev || out_table_dev == table_dev || out_input_dev == out_table_dev) { set_error("lookup_permute: outputs must not alias the inputs"); return TRH_EINVAL; }
    if (usable_rows >= ((size_t)1 << 31)) { set_error("lookup_permute: too many rows"); return TRH_EINVAL; }
    if (!usable_rows) return TRH_OK;
    TRH_ENTER(stream);
    Range range("trh_lookup_permute_dev");
    Ctx& c = ctx();
    (void)c;
    if (field == TRH_FP) return lookup_permute_t<FpParams>(input_dev, table_dev, usable_rows, out_input_dev, out_table_dev, (hipStream_t)stream);
    return lookup_permute_t<FqParams>(input_dev, table_dev, usable_rows, out_input_dev, out_table_dev, (hipStream_t)stream);
}
