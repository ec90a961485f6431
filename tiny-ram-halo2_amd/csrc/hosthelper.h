// One helper thread for host arithmetic that sits between two launches (no HIP in here: compiled and run on its own under
// -fsanitize=thread by tests/native/hosthelper_test.cpp).
//
// Why: the host side of a per-window MSM is a Horner over the window sums -- ~250 doublings per result, 70 us in hostcombine.h's 4 x 64-bit
// code -- and the results of a batch are independent.  An IPA round over the collapsed generators (ipafold.hip) waits for two of them while
// the GPU idles: with the second one on this thread the round's host turn is one Horner long instead of two.  A condition-variable wake-up
// costs 30 - 50 us, most of what there is to win, so the thread spins for a millisecond or two after every job (jobs of an opening arrive every
// 0.2 - 0.6 ms) and goes to sleep when nothing came.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace trh {

class HostHelper {
  public:
    HostHelper() : th([this] { loop(); }) {}
    ~HostHelper() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            posted.store(posted.load(std::memory_order_relaxed) + 1, std::memory_order_release);
        }
        cv.notify_all();
        th.join();
    }
    HostHelper(const HostHelper&) = delete;
    HostHelper& operator=(const HostHelper&) = delete;

    // hands f to the helper thread; one job at a time: wait() before the next start()
    void start(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(f);
            posted.store(posted.load(std::memory_order_relaxed) + 1, std::memory_order_release);
        }
        if (asleep.load(std::memory_order_acquire)) cv.notify_one();
    }
    // returns when the job handed over by the last start() has run
    void wait() {
        const unsigned want = posted.load(std::memory_order_relaxed);
        while (done.load(std::memory_order_acquire) != want) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }

  private:
    void loop() {
        unsigned seen = 0;
        for (;;) {
            for (int spin = 0; spin < 30000 && posted.load(std::memory_order_acquire) == seen; ++spin) {  // ~1 - 2 ms
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                asleep.store(true, std::memory_order_release);
                cv.wait(lk, [&] { return posted.load(std::memory_order_acquire) != seen; });
                asleep.store(false, std::memory_order_release);
                if (stop) return;
                seen = posted.load(std::memory_order_relaxed);
                f = std::move(job);
            }
            f();
            done.store(seen, std::memory_order_release);
        }
    }
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    std::atomic<unsigned> posted{0}, done{0};
    std::atomic<bool> asleep{false};
    bool stop = false;
    std::thread th;  // last: started when everything above exists
};

}  // namespace trh
