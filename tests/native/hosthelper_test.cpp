// csrc/hosthelper.h on its own (no HIP): jobs handed over back to back and after pauses long enough for the helper to fall asleep, results
// visible to the caller after wait(), clean shutdown while asleep and while spinning.  Built with -fsanitize=thread by tests/test_hostcombine.py.
#include <stdio.h>

#include <chrono>
#include <thread>
#include <vector>

#include "../../tiny-ram-halo2_amd/csrc/hosthelper.h"

int main() {
    using trh::HostHelper;
    {
        HostHelper h;
        std::vector<long> out(64, 0);
        long expect = 0;
        for (int round = 0; round < 2000; ++round) {
            const int mid = 32;
            h.start([&out, round, mid] { for (int i = mid; i < 64; ++i) out[i] += (long)round * i; });
            for (int i = 0; i < mid; ++i) out[i] += (long)round * i;
            h.wait();
            expect += round;
            if (round % 500 == 499) std::this_thread::sleep_for(std::chrono::milliseconds(30));  // the helper goes to sleep in between
        }
        for (int i = 0; i < 64; ++i)
            if (out[i] != expect * i) { printf("hosthelper: slot %d = %ld, expected %ld\n", i, out[i], expect * i); return 1; }
    }
    { HostHelper idle; }                                                                      // destroyed while spinning
    { HostHelper sleeper; std::this_thread::sleep_for(std::chrono::milliseconds(50)); }       // destroyed while asleep
    {
        HostHelper a, b;  // two contexts, two helpers
        long x = 0, y = 0;
        for (int i = 0; i < 500; ++i) {
            a.start([&x] { ++x; });
            b.start([&y] { y += 2; });
            a.wait(); b.wait();
        }
        if (x != 500 || y != 1000) { printf("hosthelper: two helpers %ld %ld\n", x, y); return 1; }
    }
    printf("hosthelper: ok\n");
    return 0;
}
