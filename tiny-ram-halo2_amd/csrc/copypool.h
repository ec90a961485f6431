// Host threads that move bytes between pageable memory and the pinned staging rings of hostio.hip (no HIP in here: the pool is
// compiled and run on its own under -fsanitize=thread by tests/native/copypool_test.cpp).
#pragma once
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace trh {

// One pool per direction (the upload side runs on the calling thread, the download side of a pipeline on its helper thread: they
// must not queue behind each other).  Heap-allocated and never destroyed: worker threads parked on a condition variable at
// process exit are harmless, a destructor joining them from a static object is not.
class CopyPool {
  public:
    explicit CopyPool(int threads) : T(threads) {
        for (int i = 0; i < T; ++i) th.emplace_back([this, i] { worker(i); });
        for (std::thread& t : th) t.detach();
    }
    // dst <- src, split over the pool and the calling thread; serialised per pool
    void copy(char* dst, const char* src, size_t bytes) {
        if (bytes < ((size_t)1 << 20) || T == 0) { memcpy(dst, src, bytes); return; }
        std::lock_guard<std::mutex> call(call_mu);
        const size_t parts = (size_t)T + 1;
        const size_t per = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu);
            job_dst = dst; job_src = src; job_bytes = bytes; job_per = per;
            pending = T;
            pending_atomic.store(T, std::memory_order_release);
            ++gen;
            gen_atomic.store(gen, std::memory_order_release);
        }
        cv_work.notify_all();
        memcpy(dst, src, per < bytes ? per : bytes);  // part 0 on the caller
        for (int spin = 0; spin < 20000 && pending_atomic.load(std::memory_order_acquire) != 0; ++spin) __builtin_ia32_pause();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }

  private:
    void worker(int id) {
        unsigned seen = 0;
        for (;;) {
            char* dst; const char* src; size_t bytes, per;
            {
                // jobs arrive every few hundred microseconds while a transfer runs: spin briefly before sleeping (a condition-variable
                // wake-up costs 30-50 us, a quarter of a slot's DMA time)
                for (int spin = 0; spin < 20000 && gen_atomic.load(std::memory_order_acquire) == seen; ++spin) __builtin_ia32_pause();
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return gen != seen; });
                seen = gen;
                dst = job_dst; src = job_src; bytes = job_bytes; per = job_per;
            }
            const size_t lo = (size_t)(id + 1) * per;
            if (lo < bytes) memcpy(dst + lo, src + lo, lo + per < bytes ? per : bytes - lo);
            {
                std::lock_guard<std::mutex> lk(mu);
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
    unsigned gen = 0;
    int pending = 0;
    std::atomic<unsigned> gen_atomic{0};
    std::atomic<int> pending_atomic{0};
};


}  // namespace trh
