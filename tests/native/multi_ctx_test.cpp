// Native (C++, no Python) test of the context layer of libtrh's C ABI on whatever GPUs the box has:
//   * a device group over {0, 0} (two logical shards on the one GPU of the test box; {0, 1, ...} when there are more):
//     range-sharded base sets give the identical point through trh_msm, trh_msm_dev and trh_best_multiexp_*
//   * two host threads, each bound to its own context (trh_ctx_create), overlap MSMs and NTTs on their own streams
//   * enqueue / finish bookkeeping: a second enqueue on a busy context is TRH_EBUSY, a finish naming another base set fails
//   * a thread that never called trh_init works (per-entry hipSetDevice), on device 1 too when the box has one
// Reference shape: one host process proving sequentially (/root/reference/src/test_utils.rs:37-54) whose rayon workers call
// best_multiexp / best_fft from arbitrary threads.   Exit code 0 iff every check passed; prints one JSON line.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "trh.h"

static int failures = 0;
#define EXPECT(cond, what) do { if (!(cond)) { ++failures; std::fprintf(stderr, "CHECK FAILED (%s:%d): %s [%s]\n", __FILE__, __LINE__, what, trh_last_error()); } } while (0)
#define OK(call) EXPECT((call) == TRH_OK, #call)

struct SplitMix {
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
};
static std::vector<uint64_t> scalars(size_t n, uint64_t seed) {  // Montgomery limbs of arbitrary residues: any value < 2^254 is a valid element
    SplitMix r{seed};
    std::vector<uint64_t> v(4 * n);
    for (size_t i = 0; i < n; ++i) { v[4 * i] = r.next(); v[4 * i + 1] = r.next(); v[4 * i + 2] = r.next(); v[4 * i + 3] = r.next() >> 2; }
    return v;
}
static bool same_point(const uint64_t* a, const uint64_t* b) { return std::memcmp(a, b, 96) == 0; }

int main() {
    if (trh_device_count() < 1) { std::fprintf(stderr, "no device\n"); return 2; }
    OK(trh_init(0));
    const size_t n = ((size_t)1 << 16) + 3;
    const std::vector<uint64_t> sc = scalars(n, 0x5eed);
    uint64_t ref[12], got[12];

    // ---- single-device reference ---------------------------------------------------------------------
    trh_bases_t single = nullptr;
    OK(trh_bases_generate(TRH_PALLAS, 0x1234567, 0x89abcdef, 0, n, &single));
    EXPECT(trh_bases_shards(single) == 1, "single-device set");
    OK(trh_msm(single, 0, sc.data(), n, 1, ref));
    std::vector<uint64_t> xy(8 * n);
    OK(trh_bases_download(single, 0, n, xy.data()));

    // ---- device group: every GPU of the box, or two logical shards on the one GPU ---------------------
    std::vector<int> devs;
    const int ndev = trh_device_count();
    if (ndev >= 2) for (int d = 0; d < ndev; ++d) devs.push_back(d); else devs = {0, 0};
    OK(trh_init_multi(devs.data(), (int)devs.size()));
    EXPECT(trh_group_size() == (int)devs.size(), "group size");
    OK(trh_set_shard_min(1000));
    trh_bases_t sharded = nullptr;
    OK(trh_bases_generate(TRH_PALLAS, 0x1234567, 0x89abcdef, 0, n, &sharded));
    EXPECT(trh_bases_shards(sharded) == (int)devs.size(), "sharded set");
    int dev_before = -1, dev_after = -1;
    EXPECT(hipGetDevice(&dev_before) == hipSuccess, "hipGetDevice");
    OK(trh_msm(sharded, 0, sc.data(), n, 1, got));
    EXPECT(same_point(ref, got), "trh_msm over the range-sharded set == single device");
    // the shard scopes are left in reverse order -- the caller's HIP device and active context are what they were
    EXPECT(hipGetDevice(&dev_after) == hipSuccess && dev_after == dev_before, "current device unchanged by a sharded MSM");
    EXPECT(trh_ctx_device(nullptr) == devs[0], "calling thread still on the default context");
    EXPECT(trh_group_peer_access() == 1 || ndev >= 2, "peer access flag");
    {   // sub-range that cuts through the shard boundary
        uint64_t a[12], b[12];
        const size_t off = n / 3, len = n / 2;
        OK(trh_msm(single, off, sc.data(), len, 1, a));
        OK(trh_msm(sharded, off, sc.data(), len, 1, b));
        EXPECT(same_point(a, b), "sharded sub-range");
    }
    {   // bases uploaded from the host are sharded the same way
        trh_bases_t up = nullptr;
        OK(trh_bases_create_pallas(xy.data(), n, &up));
        EXPECT(trh_bases_shards(up) == (int)devs.size(), "uploaded set is sharded");
        OK(trh_msm(up, 0, sc.data(), n, 1, got));
        EXPECT(same_point(ref, got), "uploaded sharded set");
        std::vector<uint64_t> back(8 * n);
        OK(trh_bases_download(up, 0, n, back.data()));
        EXPECT(back == xy, "download of a sharded set");
        trh_bases_destroy(up);
    }
    {   // device-resident scalars on the default device: handed to the shards with (peer) copies
        void* d_sc = nullptr;
        OK(trh_malloc(&d_sc, n * 32));
        OK(trh_memcpy_h2d(d_sc, sc.data(), n * 32));
        hipStream_t st;
        EXPECT(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess, "stream");
        OK(trh_msm_dev(sharded, 0, d_sc, n, 1, st, got));
        EXPECT(same_point(ref, got), "trh_msm_dev over the sharded set");
        (void)hipStreamDestroy(st);
        OK(trh_free(d_sc));
    }
    OK(trh_best_multiexp_pallas(sc.data(), xy.data(), n, got));  // the plain drop-in uses the group too
    EXPECT(same_point(ref, got), "trh_best_multiexp_pallas with a device group");
    EXPECT(trh_bases_precompute(sharded, 0) == TRH_EINVAL, "no fixed-base tables on a sharded set");
    OK(trh_set_shard_min((size_t)1 << 20));

    // ---- enqueue / finish bookkeeping -------------------------------------------------------------------
    {
        void* d_sc = nullptr;
        OK(trh_malloc(&d_sc, n * 32));
        OK(trh_memcpy_h2d(d_sc, sc.data(), n * 32));
        trh_bases_t other = nullptr;
        OK(trh_bases_generate(TRH_PALLAS, 7, 3, 0, 100, &other));
        OK(trh_msm_dev_enqueue(single, 0, d_sc, n, 1, nullptr));
        EXPECT(trh_msm_dev_enqueue(single, 0, d_sc, n, 1, nullptr) == TRH_EBUSY, "second enqueue on a busy context");
        EXPECT(trh_msm_dev_finish(other, nullptr, got) == TRH_EINVAL, "finish naming another base set");
        OK(trh_msm_dev_finish(single, nullptr, got));
        EXPECT(same_point(ref, got), "enqueue / finish");
        trh_bases_destroy(other);
        OK(trh_free(d_sc));
    }

    // ---- two threads, two contexts, own streams: MSMs beside NTTs -------------------------------------------
    {
        const size_t m = (size_t)1 << 14;
        const uint32_t log_n = 14;
        const std::vector<uint64_t> sa = scalars(m, 1), sb = scalars(m, 2);
        uint64_t ra[12], rb[12];
        OK(trh_msm(single, 0, sa.data(), m, 1, ra));
        OK(trh_msm(single, 5, sb.data(), m, 1, rb));
        trh_domain_t dom = nullptr;  // constants only: omega / omega^-1 / 2^-k for a round trip
        OK(trh_domain_create(TRH_FP, 2, log_n, &dom));
        uint64_t omega[4], omega_inv[4], ninv[4];
        OK(trh_domain_constant(dom, 0, omega)); OK(trh_domain_constant(dom, 1, omega_inv)); OK(trh_domain_constant(dom, 4, ninv));
        trh_domain_destroy(dom);
        auto worker = [&](int id, int* bad) {
            trh_ctx_t cx = nullptr;
            if (trh_ctx_create(0, &cx) != TRH_OK || trh_ctx_set_current(cx) != TRH_OK) { ++*bad; return; }
            hipStream_t st;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { ++*bad; return; }
            const std::vector<uint64_t>& s = id ? sb : sa;
            void *d_s = nullptr, *d_a = nullptr;
            if (trh_malloc(&d_s, m * 32) != TRH_OK || trh_malloc(&d_a, m * 32) != TRH_OK) { ++*bad; return; }
            if (trh_memcpy_h2d(d_s, s.data(), m * 32) != TRH_OK) ++*bad;
            std::vector<uint64_t> back(4 * m);
            for (int it = 0; it < 20; ++it) {
                uint64_t out[12];
                if (trh_msm_dev_enqueue(single, id ? 5 : 0, d_s, m, 1, st) != TRH_OK) ++*bad;
                // an NTT round trip on the same context and stream while the MSM is in flight on it
                if (trh_memcpy_h2d(d_a, s.data(), m * 32) != TRH_OK) ++*bad;
                if (trh_ntt_dev(TRH_FP, d_a, log_n, omega, 1, st) != TRH_OK) ++*bad;
                if (trh_ntt_dev(TRH_FP, d_a, log_n, omega_inv, 1, st) != TRH_OK) ++*bad;
                if (trh_field_scale_dev(TRH_FP, d_a, m, ninv, st) != TRH_OK) ++*bad;
                if (trh_msm_dev_finish(single, st, out) != TRH_OK) ++*bad;
                if (!same_point(out, id ? rb : ra)) ++*bad;
                if (trh_stream_synchronize(st) != TRH_OK) ++*bad;
                if (trh_memcpy_d2h(back.data(), d_a, m * 32) != TRH_OK) ++*bad;
                if (id == 1 && back != s) ++*bad;  // sb holds reduced residues (top limb < 2^62): the round trip is the identity
            }
            (void)trh_free(d_s); (void)trh_free(d_a);
            (void)hipStreamDestroy(st);
            (void)trh_ctx_set_current(nullptr);
            trh_ctx_destroy(cx);
        };
        int bad0 = 0, bad1 = 0;
        std::thread t0(worker, 0, &bad0), t1(worker, 1, &bad1);
        t0.join(); t1.join();
        EXPECT(bad0 == 0 && bad1 == 0, "two threads on two contexts");
    }

    // ---- a thread that never called trh_init, on the last device of the box -------------------------------------
    {
        const int dev = trh_device_count() - 1;
        int bad = 0;
        std::thread t([&] {
            trh_ctx_t cx = nullptr;
            if (trh_ctx_create(dev, &cx) != TRH_OK || trh_ctx_set_current(cx) != TRH_OK) { ++bad; return; }
            trh_bases_t b = nullptr;
            uint64_t out[12];
            if (trh_bases_generate(TRH_PALLAS, 0x1234567, 0x89abcdef, 0, n, &b) != TRH_OK) { ++bad; return; }
            if (trh_msm(b, 0, sc.data(), n, 1, out) != TRH_OK || !same_point(out, ref)) ++bad;
            trh_bases_destroy(b);
            (void)trh_ctx_set_current(nullptr);
            trh_ctx_destroy(cx);
        });
        t.join();
        EXPECT(bad == 0, "non-init thread on the last device");
        // the default context is still usable from the main thread afterwards
        OK(trh_msm(single, 0, sc.data(), n, 1, got));
        EXPECT(same_point(ref, got), "default context after the worker threads");
    }

    // ---- column-sharded commitments: one thread + one context per "GPU", each with its OWN copy of the
    // Params bases, each committing its contiguous share of 64 host columns through trh_commit_batch_host; the commitments equal
    // the single-context ones column by column (the per-column steps of create_proof shard without a collective)
    {
        const size_t cn = (size_t)1 << 12, ncols = 64;
        trh_bases_t pb = nullptr;
        OK(trh_bases_generate(TRH_VESTA, 0x1234567 + 77, 0x89abcdef + 2, 0, cn + 1, &pb));
        std::vector<std::vector<uint64_t>> cols(ncols);
        std::vector<const uint64_t*> ptrs(ncols);
        for (size_t i = 0; i < ncols; ++i) { cols[i] = scalars(cn, 0xC01 + i); ptrs[i] = cols[i].data(); }
        const std::vector<uint64_t> blinds = scalars(ncols, 0xB11D);
        std::vector<uint64_t> want(12 * ncols), gotc(12 * ncols, 0);
        for (size_t i = 0; i < ncols; ++i) {
            std::vector<uint64_t> scb(cols[i]);
            scb.insert(scb.end(), blinds.begin() + 4 * i, blinds.begin() + 4 * i + 4);
            OK(trh_msm(pb, 0, scb.data(), cn + 1, 1, want.data() + 12 * i));
        }
        OK(trh_commit_batch_host(pb, ptrs.data(), cn, ncols, blinds.data(), gotc.data()));
        EXPECT(gotc == want, "trh_commit_batch_host == per-column trh_msm");
        const size_t G = devs.size();
        std::vector<int> bad(G, 0);
        std::fill(gotc.begin(), gotc.end(), 0);
        std::vector<std::thread> th;
        for (size_t gi = 0; gi < G; ++gi)
            th.emplace_back([&, gi] {
                trh_ctx_t cx = nullptr;
                if (trh_ctx_create(devs[gi], &cx) != TRH_OK || trh_ctx_set_current(cx) != TRH_OK) { ++bad[gi]; return; }
                trh_bases_t mine = nullptr;
                if (trh_bases_generate(TRH_VESTA, 0x1234567 + 77, 0x89abcdef + 2, 0, cn + 1, &mine) != TRH_OK) { ++bad[gi]; return; }
                const size_t per = (ncols + G - 1) / G, lo = gi * per, hi = lo + per < ncols ? lo + per : ncols;
                if (hi > lo && trh_commit_batch_host(mine, ptrs.data() + lo, cn, hi - lo, blinds.data() + 4 * lo, gotc.data() + 12 * lo) != TRH_OK) ++bad[gi];
                trh_bases_destroy(mine);
                (void)trh_ctx_set_current(nullptr);
                trh_ctx_destroy(cx);
            });
        for (std::thread& t : th) t.join();
        int badsum = 0;
        for (int b : bad) badsum += b;
        EXPECT(badsum == 0 && gotc == want, "column-sharded commitments: one thread + context + Params copy per device");
        trh_bases_destroy(pb);
    }

    trh_bases_destroy(sharded);
    trh_bases_destroy(single);
    trh_shutdown();
    std::printf("{\"test\": \"multi_ctx\", \"devices\": %d, \"group\": %zu, \"checks_failed\": %d}\n", ndev, devs.size(), failures);
    return failures ? 1 : 0;
}
