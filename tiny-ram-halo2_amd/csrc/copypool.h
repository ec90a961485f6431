// Host threads that move bytes between pageable memory and the pinned staging rings of hostio.hip (no HIP in here: the pool is
// compiled and run on its own under -fsanitize=thread by tests/native/copypool_test.cpp).
#pragma once
#include <errno.h>
#include <stdint.h>
#include <string.h>
#if defined(__linux__)
#include <sys/mman.h>
#endif

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define TRH_HAVE_NT_COPY 1
#endif

namespace trh {

#ifdef TRH_HAVE_NT_COPY
// dst <- src with non-temporal stores: a slot-sized copy with ordinary stores reads every destination line before it overwrites it
// (read-for-ownership), i.e. moves 3 bytes of DRAM traffic per byte copied instead of 2 -- on a host whose memory also feeds / drains the
// DMA engine at link rate that traffic is what the link waits for.  glibc switches to streaming stores only above ~3/4 of the shared
// cache per call; a pool thread's share of a 16 MiB slot is below that.
__attribute__((target("avx2"))) inline void nt_copy_avx2(char* dst, const char* src, size_t n) {
    size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
    if (head > n) head = n;
    memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    const size_t blocks = n / 128;
    for (size_t i = 0; i < blocks; ++i) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)src), b = _mm256_loadu_si256((const __m256i*)(src + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i*)(src + 64)), d = _mm256_loadu_si256((const __m256i*)(src + 96));
        _mm256_stream_si256((__m256i*)dst, a); _mm256_stream_si256((__m256i*)(dst + 32), b);
        _mm256_stream_si256((__m256i*)(dst + 64), c); _mm256_stream_si256((__m256i*)(dst + 96), d);
        src += 128; dst += 128;
    }
    _mm_sfence();
    memcpy(dst, src, n - blocks * 128);
}
#endif
// one thread's share of a copy; nt: streaming stores where the CPU has them (the caller's choice per direction)
inline void copy_bytes(char* dst, const char* src, size_t n, bool nt) {
#ifdef TRH_HAVE_NT_COPY
    static const bool have = __builtin_cpu_supports("avx2");
    if (nt && have && n >= ((size_t)64 << 10)) { nt_copy_avx2(dst, src, n); return; }
#endif
    memcpy(dst, src, n);
}

// true when every byte of [p, p + bytes) is zero.  Data that is not zero answers after the first word; a zero range costs one read pass
// (no write), which is what the zero-padded vectors of coeff_to_extended are for 7/8 of their length.
inline bool all_zero(const char* p, size_t bytes) {
    size_t i = 0;
    for (; i < bytes && ((uintptr_t)(p + i) & 7); ++i) if (p[i]) return false;
    const uint64_t* w = (const uint64_t*)(p + i);
    const size_t nw = (bytes - i) / 8;
    size_t k = 0;
    for (; k + 8 <= nw; k += 8)
        if (w[k] | w[k + 1] | w[k + 2] | w[k + 3] | w[k + 4] | w[k + 5] | w[k + 6] | w[k + 7]) return false;
    for (; k < nw; ++k) if (w[k]) return false;
    for (i += nw * 8; i < bytes; ++i) if (p[i]) return false;
    return true;
}

// One pool per direction and context (the upload side runs on the calling thread, the download side of a pipeline on its helper
// thread: they must not queue behind each other; two contexts -- two GPUs -- must not queue behind each other either).  A context's
// pools die with it (the destructor parks no thread: it stops and joins them); the process-wide pools of tests are heap-allocated
// and never destroyed (a destructor joining threads from a static object at process exit is not harmless).
class CopyPool {
  public:
    explicit CopyPool(int threads) : T(threads < 0 ? 0 : threads > 62 ? 62 : threads) {  // the zero mask has a bit per part
        for (int i = 0; i < T; ++i) th.emplace_back([this, i] { worker(i); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            ++gen;
            gen_atomic.store(gen, std::memory_order_release);
        }
        cv_work.notify_all();
        for (std::thread& t : th) t.join();
    }
    CopyPool(const CopyPool&) = delete;
    CopyPool& operator=(const CopyPool&) = delete;

    // dst <- src, split over the pool and the calling thread; serialised per pool.
    // detect_zero: a source that is zero throughout is NOT copied and the call returns true (the caller replaces the transfer by a
    // device-side memset); parts of a mixed source that are zero are cleared in dst, the call returns false.
    // nt: streaming stores (see nt_copy_avx2)
    // scan_only (with detect_zero): nothing is written at all -- the call answers "is the source zero throughout?" with the pool's threads
    bool copy(char* dst, const char* src, size_t bytes, bool detect_zero = false, bool nt = false, bool scan_only = false) {
        if (bytes < ((size_t)1 << 20) || T == 0) {
            if (detect_zero && bytes && all_zero(src, bytes)) return true;
            if (!scan_only) copy_bytes(dst, src, bytes, nt);
            return false;
        }
        std::lock_guard<std::mutex> call(call_mu);
        const size_t parts = (size_t)T + 1;
        const size_t per = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu);
            job_dst = dst; job_src = src; job_bytes = bytes; job_per = per; job_detect = detect_zero; job_nt = nt; job_scan = scan_only;
            zero_mask = 0;
            pending = T;
            pending_atomic.store(T, std::memory_order_release);
            ++gen;
            gen_atomic.store(gen, std::memory_order_release);
        }
        cv_work.notify_all();
        const bool z0 = part(0, dst, src, bytes, per, detect_zero, nt, scan_only);  // part 0 on the caller
        for (int spin = 0; spin < 20000 && pending_atomic.load(std::memory_order_acquire) != 0; ++spin) __builtin_ia32_pause();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        if (!detect_zero) return false;
        uint64_t mask = zero_mask | (z0 ? 1u : 0u);
        size_t live = 0;
        for (size_t k = 0; k < parts; ++k) if (k * per < bytes) ++live;
        if (mask == (live >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << live) - 1))) return true;  // zero throughout: dst untouched
        if (scan_only) return false;
        for (size_t k = 0; k < live; ++k)  // a mixed slot (the boundary between the data and its padding): the skipped parts are cleared
            if (mask >> k & 1) memset(dst + k * per, 0, (k + 1) * per < bytes ? per : bytes - k * per);
        return false;
    }

  private:
    // part k of the job; returns true when detect was asked and the part is zero (and was therefore NOT written)
    static bool part(size_t k, char* dst, const char* src, size_t bytes, size_t per, bool detect, bool nt, bool scan) {
        const size_t lo = k * per;
        if (lo >= bytes) return false;
        const size_t len = lo + per < bytes ? per : bytes - lo;
        if (detect && all_zero(src + lo, len)) return true;
        if (!scan) copy_bytes(dst + lo, src + lo, len, nt);
        return false;
    }
    void worker(int id) {
        unsigned seen = 0;
        for (;;) {
            char* dst; const char* src; size_t bytes, per; bool detect, nt, scan;
            {
                // jobs arrive every few hundred microseconds while a transfer runs: spin briefly before sleeping (a condition-variable
                // wake-up costs 30-50 us, a quarter of a slot's DMA time)
                for (int spin = 0; spin < 20000 && gen_atomic.load(std::memory_order_acquire) == seen; ++spin) __builtin_ia32_pause();
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return gen != seen; });
                if (stop) return;
                seen = gen;
                dst = job_dst; src = job_src; bytes = job_bytes; per = job_per; detect = job_detect; nt = job_nt; scan = job_scan;
            }
            const bool z = part((size_t)id + 1, dst, src, bytes, per, detect, nt, scan);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (z) zero_mask |= (uint64_t)1 << (id + 1);
                pending_atomic.store(pending - 1, std::memory_order_release);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    const int T;
    std::vector<std::thread> th;
    std::mutex mu, call_mu;
    std::condition_variable cv_work, cv_done;
    char* job_dst = nullptr; const char* job_src = nullptr; size_t job_bytes = 0, job_per = 0;
    bool job_detect = false, job_nt = false, job_scan = false, stop = false;
    uint64_t zero_mask = 0;
    unsigned gen = 0;
    int pending = 0;
    std::atomic<unsigned> gen_atomic{0};
    std::atomic<int> pending_atomic{0};
};


// Chunk sizes of ONE transfer through a slot ring.  Full slots in the middle; a short head (2 MiB, then 6) when the transfer's first
// bytes gate the pipeline (an upload: nothing moves until the first chunk sits in pinned memory) and a short tail (6, then 2) when its
// last bytes do (a download: the caller waits for the copy out of the last chunk; an upload whose host copies are the slower side).
// The fill / drain of a 16 MiB ring cost 0.25 ms each on a 128 MiB best_fft; uniformly small slots lose to the per-slot hand-over.
// No chunk is ever larger than a slot (TRH_STAGE_SLOT_MB may be as small as 1: slots no larger than the short chunk are not graded).
// tiny_tail (bytes, 0 = none): a last chunk of that size in front of everything else that is cut off the end -- for transfers whose zero chunks are
// elided and whose last bytes are NOT zero (a witness column: zero padding, then a few blinding rows): the padding in front of it then elides whole.
inline void chunk_plan(size_t bytes, size_t slot, bool head, bool tail, std::vector<size_t>& out, size_t tiny_tail = 0) {
    out.clear();
    if (!slot) return;
    const size_t small = (size_t)2 << 20, mid = (size_t)6 << 20;
    size_t left = bytes;
    std::vector<size_t> back;
    if (tiny_tail && tiny_tail < slot && left > 4 * tiny_tail) { back.push_back(tiny_tail); left -= tiny_tail; }
    if (slot > small) {
        if (head && left > 2 * small) { out.push_back(small); left -= small; if (slot > mid && left > mid + small) { out.push_back(mid); left -= mid; } }
        if (tail && left > 2 * small) { back.push_back(small); left -= small; if (slot > mid && left > mid + small) { back.push_back(mid); left -= mid; } }
    }
    while (left) { const size_t cur = left < slot ? left : slot; out.push_back(cur); left -= cur; }
    for (size_t i = back.size(); i-- > 0;) out.push_back(back[i]);
}

}  // namespace trh
