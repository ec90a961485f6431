// FETCH_SIZE calibration for the MSM accumulation's access pattern.
//
// MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read; other access widths are
// uncalibrated.  msm_accumulate_seg_kernel does not stream: every lane gathers 80 of the 128 bytes of a RANDOM 128-byte base record with
// 16-byte loads.  This probe issues exactly that pattern over a table far larger than L2 + Infinity Cache with a byte count known by
// construction (every record is read exactly once per launch: record = (thread * odd) mod 2^log_records, a bijection), next to the streaming
// pattern the guide calibrated, so that FETCH_SIZE read back under `rocprofv3 --pmc FETCH_SIZE` gives the factor for each:
//     stream16      64 lanes x 16 B consecutive                          bytes = table
//     gather80      5 x 16 B at offsets 0..79 of a random record          requested = 80 B / record, lines touched = 128 B / record
//     gather128     8 x 16 B: the whole random record                     bytes = 128 B / record
//     gather64lo    4 x 16 B at offsets 0..63                             requested = 64 B / record: does a half-used 128-B line cost 64 or 128?
// usage: tools/gather_probe [log_records = 24] [reps = 3]   (under rocprofv3 --kernel-trace --pmc FETCH_SIZE; tools/pmc_gather.sh)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) fill_kernel(uint4* t, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) t[i] = make_uint4((uint32_t)i, (uint32_t)(i >> 7), 0x9e3779b9u, (uint32_t)(i * 2654435761u));
}
__global__ void __launch_bounds__(256) stream16_kernel(const uint4* __restrict__ t, size_t n16, uint32_t* __restrict__ sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = t[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;  // never true in practice: keeps the loads alive
}
// a bijection of [0, 2^k): odd multiplications and right xor-shifts are each invertible mod 2^k
__host__ __device__ inline size_t scatter_index(size_t i, size_t mask) {
    size_t r = (i * 0x9E3779B97F4A7C15ull) & mask;
    r ^= r >> 11;
    r = (r * 0xD6E8FEB86659FD93ull) & mask;
    r ^= r >> 13;
    return r;
}
template <int FIRST, int COUNT>
__global__ void __launch_bounds__(256) gather_kernel(const uint4* __restrict__ t, uint32_t log_records, uint32_t* __restrict__ sink) {
    const size_t records = (size_t)1 << log_records;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < records; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = scatter_index(i, records - 1);
        const uint4* p = t + r * 8 + FIRST;
#pragma unroll
        for (int k = 0; k < COUNT; ++k) { const uint4 v = p[k]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
    const uint32_t log_records = argc > 1 ? (uint32_t)atoi(argv[1]) : 24;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t records = (size_t)1 << log_records, bytes = records * 128, n16 = bytes / 16;
    uint4* t; uint32_t* sink;
    CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMemset(sink, 0, 64));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, t, n16);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // "every record exactly once per launch" as a checked statement
    {
        uint8_t* seen = (uint8_t*)calloc(records / 8 + 1, 1);
        size_t distinct = 0;
        for (size_t i = 0; i < records; ++i) { const size_t r = scatter_index(i, records - 1); if (!(seen[r >> 3] >> (r & 7) & 1)) { seen[r >> 3] |= (uint8_t)(1u << (r & 7)); ++distinct; } }
        printf("# table %zu records x 128 B = %.3f GB; distinct records touched per gather launch: %zu (%.4f of all)\n", records, bytes / 1e9, distinct, (double)distinct / records);
        free(seen);
    }
    auto time_it = [&](const char* name, auto launch, double requested, double lines) {
        launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("%-12s %8.3f ms   requested %.4e B (%.0f GB/s)   whole lines %.4e B (%.0f GB/s)\n", name, ms, requested, requested / ms / 1e6, lines, lines / ms / 1e6);
    };
    const dim3 grid(256 * 16), wg(256);
    time_it("stream16", [&] { hipLaunchKernelGGL(stream16_kernel, grid, wg, 0, 0, t, n16, sink); }, (double)bytes, (double)bytes);
    time_it("gather80", [&] { hipLaunchKernelGGL((gather_kernel<0, 5>), grid, wg, 0, 0, t, log_records, sink); }, records * 80.0, records * 128.0);
    time_it("gather128", [&] { hipLaunchKernelGGL((gather_kernel<0, 8>), grid, wg, 0, 0, t, log_records, sink); }, records * 128.0, records * 128.0);
    time_it("gather64lo", [&] { hipLaunchKernelGGL((gather_kernel<0, 4>), grid, wg, 0, 0, t, log_records, sink); }, records * 64.0, records * 128.0);
    CHECK(hipFree(t)); CHECK(hipFree(sink));
    return 0;
}
