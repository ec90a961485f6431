// Host <-> device transfer rates on the box, to price the host-pointer entry points (trh_best_fft_*, trh_best_multiexp_*, trh_msm):
//   pageable hipMemcpy, pinned hipMemcpyAsync, hipHostRegister / Unregister cost, and a staged copy through a pinned ring with T
//   helper threads doing the pageable <-> pinned memcpy (what csrc/stage.h does).
// build: hipcc -O2 -std=c++17 tools/pcie_probe.cpp -o tools/pcie_probe -pthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_memcpy(char* dst, const char* src, size_t bytes, int T) {
    if (T <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    size_t per = (bytes / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo >= hi) break;
        th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
    size_t mb = argc > 1 ? atol(argv[1]) : 512;
    size_t bytes = mb << 20;
    CK(hipSetDevice(0));
    char* page = (char*)aligned_alloc(4096, bytes);
    memset(page, 1, bytes);
    char* page2 = (char*)aligned_alloc(4096, bytes);
    memset(page2, 2, bytes);
    char *pin = nullptr, *dev = nullptr, *dev2 = nullptr;
    CK(hipHostMalloc((void**)&pin, bytes, hipHostMallocDefault));
    memset(pin, 3, bytes);
    CK(hipMalloc((void**)&dev, bytes));
    CK(hipMalloc((void**)&dev2, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    printf("cores %u, buffer %zu MiB\n", std::thread::hardware_concurrency(), mb);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipMemcpy(dev, page, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        CK(hipMemcpy(page2, dev, bytes, hipMemcpyDeviceToHost));
        double t2 = now();
        printf("pageable hipMemcpy      H2D %6.2f GB/s   D2H %6.2f GB/s\n", bytes / (t1 - t0) / 1e9, bytes / (t2 - t1) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        double t1 = now();
        CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1));
        double t2 = now();
        printf("pinned hipMemcpyAsync   H2D %6.2f GB/s   D2H %6.2f GB/s\n", bytes / (t1 - t0) / 1e9, bytes / (t2 - t1) / 1e9);
    }
    {   // both directions at once
        double t0 = now();
        CK(hipMemcpyAsync(dev, pin, bytes / 2, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(pin + bytes / 2, dev2, bytes / 2, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        double t1 = now();
        printf("pinned duplex           %6.2f GB/s per direction (%.2f total)\n", bytes / 2 / (t1 - t0) / 1e9, bytes / (t1 - t0) / 1e9);
    }
    for (size_t sz : {(size_t)64 << 10, (size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20}) {
        if (sz > bytes) continue;
        int reps = 20;
        double t0 = now();
        for (int i = 0; i < reps; ++i) { CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
        double t1 = now();
        for (int i = 0; i < reps; ++i) { CK(hipMemcpy(dev, page, sz, hipMemcpyHostToDevice)); }
        double t2 = now();
        for (int i = 0; i < reps; ++i) { CK(hipMemcpy(page2, dev, sz, hipMemcpyDeviceToHost)); }
        double t3 = now();
        printf("size %8zu KiB: pinned H2D %7.1f us (%5.1f GB/s)  pageable H2D %7.1f us (%5.1f GB/s)  pageable D2H %7.1f us (%5.1f GB/s)\n", sz >> 10,
               (t1 - t0) / reps * 1e6, sz * reps / (t1 - t0) / 1e9, (t2 - t1) / reps * 1e6, sz * reps / (t2 - t1) / 1e9, (t3 - t2) / reps * 1e6, sz * reps / (t3 - t2) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipHostRegister(page, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dev, page, bytes, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        double t2 = now();
        CK(hipHostUnregister(page));
        double t3 = now();
        printf("hipHostRegister %6.2f GB/s (%.2f ms)  copy %6.2f GB/s  unregister %.2f ms\n", bytes / (t1 - t0) / 1e9, (t1 - t0) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
    }
    for (int T : {1, 2, 4, 8, 16}) {
        double t0 = now();
        par_memcpy(pin, page, bytes, T);
        double t1 = now();
        par_memcpy(page2, pin, bytes, T);
        double t2 = now();
        printf("host memcpy %2d threads   pageable->pinned %6.2f GB/s   pinned->pageable %6.2f GB/s\n", T, bytes / (t1 - t0) / 1e9, bytes / (t2 - t1) / 1e9);
    }
    {   // the prover's case: every column is a DIFFERENT pageable buffer, seen by the runtime for the first time
        const int NB = 16;
        const size_t sz = (size_t)64 << 20;
        std::vector<char*> bufs(NB);
        for (int i = 0; i < NB; ++i) { bufs[i] = (char*)aligned_alloc(4096, sz); memset(bufs[i], i, sz); }
        double t0 = now();
        for (int i = 0; i < NB; ++i) CK(hipMemcpy(dev, bufs[i], sz, hipMemcpyHostToDevice));
        double t1 = now();
        for (int i = 0; i < NB; ++i) CK(hipMemcpy(dev, bufs[i], sz, hipMemcpyHostToDevice));
        double t2 = now();
        printf("pageable H2D, 16 distinct 64 MiB buffers: first sight %6.2f GB/s, second pass %6.2f GB/s\n", NB * sz / (t1 - t0) / 1e9, NB * sz / (t2 - t1) / 1e9);
        std::vector<char*> outs(NB);
        for (int i = 0; i < NB; ++i) { outs[i] = (char*)aligned_alloc(4096, sz); }
        t0 = now();
        for (int i = 0; i < NB; ++i) CK(hipMemcpy(outs[i], dev, sz, hipMemcpyDeviceToHost));   // untouched pages: faults + pinning
        t1 = now();
        for (int i = 0; i < NB; ++i) CK(hipMemcpy(outs[i], dev, sz, hipMemcpyDeviceToHost));
        t2 = now();
        printf("pageable D2H, 16 distinct 64 MiB buffers: first sight (untouched pages) %6.2f GB/s, second pass %6.2f GB/s\n", NB * sz / (t1 - t0) / 1e9, NB * sz / (t2 - t1) / 1e9);
        // does hipMemcpyAsync return before a pageable copy is done?
        t0 = now();
        CK(hipMemcpyAsync(dev, bufs[0], sz, hipMemcpyHostToDevice, s1));
        t1 = now();
        CK(hipStreamSynchronize(s1));
        t2 = now();
        printf("hipMemcpyAsync pageable 64 MiB H2D: call returns after %.3f ms, done after %.3f ms\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3);
        t0 = now();
        CK(hipMemcpyAsync(outs[0], dev, sz, hipMemcpyDeviceToHost, s1));
        t1 = now();
        CK(hipStreamSynchronize(s1));
        t2 = now();
        printf("hipMemcpyAsync pageable 64 MiB D2H: call returns after %.3f ms, done after %.3f ms\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3);
        t0 = now();
        CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s1));
        t1 = now();
        CK(hipStreamSynchronize(s1));
        t2 = now();
        printf("hipMemcpyAsync pinned   64 MiB H2D: call returns after %.3f ms, done after %.3f ms\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3);
        // staged through the pinned buffer with one memcpy thread (what a fresh pageable buffer costs without touching the runtime's pinning)
        t0 = now();
        for (int i = 0; i < NB; ++i) { memcpy(pin, bufs[i], sz); CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
        t1 = now();
        printf("memcpy to pinned + pinned H2D, serial, 1 thread: %6.2f GB/s\n", NB * sz / (t1 - t0) / 1e9);
        // two directions at once, separate device buffers, large transfers
        t0 = now();
        CK(hipMemcpyAsync(dev, pin, bytes / 2, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(pin + bytes / 2, dev2, bytes / 2, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        t1 = now();
        CK(hipMemcpyAsync(dev, pin, bytes / 2, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        t2 = now();
        printf("duplex again: %.2f ms for %zu MiB each way at once; one way alone %.2f ms\n", (t1 - t0) * 1e3, bytes >> 21, (t2 - t1) * 1e3);
    }
    if (argc > 2) return 0;
    // staged pipeline: slots of S MiB, T copy threads
    for (size_t slot_mb : {(size_t)2, (size_t)8, (size_t)32}) {
        for (int T : {1, 4, 8}) {
            const size_t S = slot_mb << 20;
            const int NS = 4;
            hipEvent_t ev[NS];
            for (int i = 0; i < NS; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            double t0 = now();
            size_t off = 0;
            int k = 0;
            while (off < bytes) {
                size_t c = bytes - off < S ? bytes - off : S;
                int sl = k % NS;
                if (k >= NS) CK(hipEventSynchronize(ev[sl]));
                par_memcpy(pin + (size_t)sl * S, page + off, c, T);
                CK(hipMemcpyAsync(dev + off, pin + (size_t)sl * S, c, hipMemcpyHostToDevice, s1));
                CK(hipEventRecord(ev[sl], s1));
                off += c;
                ++k;
            }
            CK(hipStreamSynchronize(s1));
            double t1 = now();
            // D2H: issue copies ahead, drain with the threads
            off = 0; k = 0;
            size_t issued = 0; int ki = 0;
            while (off < bytes) {
                while (issued < bytes && ki < k + NS) {
                    size_t c = bytes - issued < S ? bytes - issued : S;
                    CK(hipMemcpyAsync(pin + (size_t)(ki % NS) * S, dev + issued, c, hipMemcpyDeviceToHost, s2));
                    CK(hipEventRecord(ev[ki % NS], s2));
                    issued += c; ++ki;
                }
                size_t c = bytes - off < S ? bytes - off : S;
                CK(hipEventSynchronize(ev[k % NS]));
                par_memcpy(page2 + off, pin + (size_t)(k % NS) * S, c, T);
                off += c; ++k;
            }
            double t2 = now();
            printf("staged ring slot %2zu MiB x4, %d threads: H2D %6.2f GB/s  D2H %6.2f GB/s\n", slot_mb, T, bytes / (t1 - t0) / 1e9, bytes / (t2 - t1) / 1e9);
            for (int i = 0; i < NS; ++i) CK(hipEventDestroy(ev[i]));
        }
    }
    return 0;
}
