// The copy pool of the host-pointer staging path (csrc/copypool.h) on its own: concurrent callers on two pools (the upload side and a
// pipeline's download helper), sizes around the slicing boundaries, content checked byte for byte.  Built with -fsanitize=thread (and
// address) by tests/test_hostcombine.py: the hand-over through the generation counter / pending count must be race-free.
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "../../tiny-ram-halo2_amd/csrc/copypool.h"

int main() {
    trh::CopyPool* up = new trh::CopyPool(3);
    trh::CopyPool* down = new trh::CopyPool(2);
    trh::CopyPool* none = new trh::CopyPool(0);
    int bad = 0;
    auto run = [&](trh::CopyPool* pool, unsigned seed, int* fails) {
        const size_t sizes[] = {0, 1, 4095, 4096, (1u << 20) - 1, 1u << 20, (1u << 20) + 1, (3u << 20) + 12345, 8u << 20, (8u << 20) + 4097};
        for (int rep = 0; rep < 6; ++rep)
            for (size_t sz : sizes) {
                std::vector<unsigned char> src(sz + 64), dst(sz + 64, 0xEE);
                for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)((i * 2654435761u + seed + rep) >> 13);
                pool->copy((char*)dst.data() + 32, (const char*)src.data() + 32, sz);
                if (sz && memcmp(dst.data() + 32, src.data() + 32, sz) != 0) ++*fails;
                for (int g = 0; g < 32; ++g) if (dst[g] != 0xEE || dst[32 + sz + g] != 0xEE) { ++*fails; break; }  // nothing outside the range
            }
    };
    int f[4] = {0, 0, 0, 0};
    std::thread a(run, up, 1u, &f[0]), b(run, up, 2u, &f[1]), c(run, down, 3u, &f[2]);  // two callers share one pool, a third uses the other
    run(none, 4u, &f[3]);
    a.join(); b.join(); c.join();
    for (int v : f) bad += v;
    // zero detection (the zero-padded vectors of coeff_to_extended): an all-zero source is reported and dst is left alone; a source
    // whose tail is zero is copied with the skipped parts cleared; a single non-zero byte anywhere defeats the detection
    for (trh::CopyPool* pool : {up, none}) {
        const size_t sizes[] = {1, 4096, (1u << 20) - 1, (1u << 20) + 1, (5u << 20) + 777, 16u << 20};
        for (size_t sz : sizes) {
            std::vector<unsigned char> src(sz, 0), dst(sz, 0xEE);
            if (!pool->copy(nullptr, (const char*)src.data(), sz, true, false, true)) ++bad;                    // scan only: zero, no destination needed
            if (!pool->copy((char*)dst.data(), (const char*)src.data(), sz, true)) ++bad;                      // zero throughout
            for (size_t i = 0; i < sz; i += 997) if (dst[i] != 0xEE) { ++bad; break; }                           // ... and dst untouched
            for (size_t pos : {(size_t)0, sz / 2, sz - 1}) {
                std::fill(src.begin(), src.end(), 0); std::fill(dst.begin(), dst.end(), 0xEE);
                src[pos] = 7;
                if (pool->copy(nullptr, (const char*)src.data(), sz, true, false, true)) ++bad;                 // scan only: not zero, nothing written
                if (pool->copy((char*)dst.data(), (const char*)src.data(), sz, true)) ++bad;
                if (memcmp(dst.data(), src.data(), sz) != 0) ++bad;                                                // the zero parts were cleared
            }
            std::fill(dst.begin(), dst.end(), 0xEE);
            for (size_t i = 0; i < sz / 8 + 1 && i < sz; ++i) src[i] = (unsigned char)(i * 31 + 1);               // data, then padding
            if (pool->copy((char*)dst.data(), (const char*)src.data(), sz, true)) ++bad;
            if (memcmp(dst.data(), src.data(), sz) != 0) ++bad;
        }
    }
    // streaming-store copies (every alignment of source and destination, sizes around the 128-byte blocks)
    for (trh::CopyPool* pool : {up, none}) {
        for (size_t sz : {(size_t)100, (size_t)65536, (size_t)65536 + 127, (size_t)(3u << 20) + 77}) {
            for (int ao : {0, 1, 31, 33}) {
                std::vector<unsigned char> src(sz + 64), dst(sz + 128, 0xEE);
                for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)(i * 131 + sz);
                pool->copy((char*)dst.data() + 32 + ao, (const char*)src.data() + (ao ^ 1), sz, false, true);
                if (memcmp(dst.data() + 32 + ao, src.data() + (ao ^ 1), sz) != 0) ++bad;
                for (int g = 0; g < 32 + ao; ++g) if (dst[g] != 0xEE) { ++bad; break; }
                for (size_t g = 32 + ao + sz; g < dst.size(); ++g) if (dst[g] != 0xEE) { ++bad; break; }
            }
        }
    }
    // chunk_plan: every chunk fits a ring slot for every slot size TRH_STAGE_SLOT_MB admits (1 ... 256 MiB), the chunks
    // add up to the transfer, and a graded head / tail only appears where it is shorter than a slot
    {
        const size_t MiB = (size_t)1 << 20;
        const size_t sizes[] = {1, MiB - 1, MiB, 2 * MiB, 4 * MiB, 4 * MiB + 1, 5 * MiB + 3, 8 * MiB, 14 * MiB + 5, 16 * MiB, 33 * MiB + 7, 128 * MiB, 600 * MiB + 11};
        std::vector<size_t> plan;
        for (size_t slot_mb = 1; slot_mb <= 256; ++slot_mb)
            for (size_t bytes : sizes)
                for (int ht = 0; ht < 4; ++ht) {
                    trh::chunk_plan(bytes, slot_mb * MiB, (ht & 1) != 0, (ht & 2) != 0, plan);
                    size_t sum = 0;
                    for (size_t cur : plan) { sum += cur; if (!cur || cur > slot_mb * MiB) ++bad; }
                    if (sum != bytes) ++bad;
                }
        trh::chunk_plan(0, 16 * MiB, true, true, plan);
        if (!plan.empty()) ++bad;
        trh::chunk_plan(128 * MiB, 16 * MiB, true, true, plan);  // the default ring: 2, 6, full slots, 6, 2
        if (plan.size() < 5 || plan[0] != 2 * MiB || plan[1] != 6 * MiB || plan[plan.size() - 2] != 6 * MiB || plan.back() != 2 * MiB) ++bad;
    }
    delete up; delete down; delete none;  // the destructor stops and joins the workers (per-context pools die with their context)
    std::printf(bad ? "copypool: FAILED (%d)\n" : "copypool: ok\n", bad);
    return bad ? 1 : 0;
}
