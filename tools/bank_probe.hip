// Do VGPR bank conflicts cost v_mad_i64_i32 issue cycles on gfx950?  (Round 4: msm_accumulate_seg_kernel runs at 4.06 cycles per VALU instruction
// against 3.69 of the per-class issue model; instruction fetch and exposed memory latency are ruled out by counters.)  Pure-register loops with
// EXPLICIT physical registers (bank = register number mod 4), 8 independent accumulators per iteration:
//   free      acc pair in banks (0,1), the two 32-bit sources in banks 2 and 3
//   ab_same   both 32-bit sources in bank 2
//   a_on_acc  one source in the accumulator's low bank
//   all_same  both sources in the accumulator's low bank
// build: hipcc -O3 --offload-arch=gfx950 tools/bank_probe.hip -o tools/bank_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 16384;

// accumulators v[8:9] v[12:13] ... v[36:37] (banks 0,1); sources picked per variant
#define ROW(A, B) \
    "v_mad_i64_i32 v[8:9], vcc, " A ", " B ", v[8:9]\n" \
    "v_mad_i64_i32 v[12:13], vcc, " A ", " B ", v[12:13]\n" \
    "v_mad_i64_i32 v[16:17], vcc, " A ", " B ", v[16:17]\n" \
    "v_mad_i64_i32 v[20:21], vcc, " A ", " B ", v[20:21]\n" \
    "v_mad_i64_i32 v[24:25], vcc, " A ", " B ", v[24:25]\n" \
    "v_mad_i64_i32 v[28:29], vcc, " A ", " B ", v[28:29]\n" \
    "v_mad_i64_i32 v[32:33], vcc, " A ", " B ", v[32:33]\n" \
    "v_mad_i64_i32 v[36:37], vcc, " A ", " B ", v[36:37]\n"
#define CLOBBERS "vcc", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", "v36", "v37", "v42", "v43", "v46", "v40", "v44"
#define KERNEL(NAME, A, B)                                                                                       \
    __global__ void __launch_bounds__(256) NAME(uint64_t* out, int a, int b) {                                   \
        asm volatile("v_mov_b32 v42, %0\n v_mov_b32 v43, %1\n v_mov_b32 v46, %1\n v_mov_b32 v40, %0\n v_mov_b32 v44, %1\n" \
                     "v_mov_b32 v8, 1\n v_mov_b32 v9, 0\n v_mov_b32 v12, 2\n v_mov_b32 v13, 0\n v_mov_b32 v16, 3\n v_mov_b32 v17, 0\n v_mov_b32 v20, 4\n v_mov_b32 v21, 0\n" \
                     "v_mov_b32 v24, 5\n v_mov_b32 v25, 0\n v_mov_b32 v28, 6\n v_mov_b32 v29, 0\n v_mov_b32 v32, 7\n v_mov_b32 v33, 0\n v_mov_b32 v36, 8\n v_mov_b32 v37, 0\n" \
                     :: "v"(a + (int)threadIdx.x), "v"(b) : CLOBBERS);                                            \
        for (int i = 0; i < ITERS; ++i) asm volatile(ROW(A, B) ::: CLOBBERS);                                    \
        uint32_t lo;                                                                                             \
        asm volatile("v_xor_b32 %0, v8, v36" : "=v"(lo) :: CLOBBERS);                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = lo;                                                         \
    }
// v42: bank 2, v43: bank 3, v46: bank 2, v40: bank 0, v44: bank 0
KERNEL(k_free, "v42", "v43")
KERNEL(k_ab_same, "v42", "v46")
KERNEL(k_a_on_acc, "v40", "v43")
KERNEL(k_all_same, "v40", "v44")

template <class K> double run(K kern, int blocks, uint64_t* d) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern(blocks, d);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) kern(blocks, d);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}
int main() {
    uint64_t* d;
    CK(hipMalloc(&d, 8 * 4096 * 256));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    for (int wps : {2, 3, 4}) {
        const int blocks = prop.multiProcessorCount * wps;
        const double f = run([](int b, uint64_t* o) { hipLaunchKernelGGL(k_free, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d);
        const double s = run([](int b, uint64_t* o) { hipLaunchKernelGGL(k_ab_same, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d);
        const double a = run([](int b, uint64_t* o) { hipLaunchKernelGGL(k_a_on_acc, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d);
        const double l = run([](int b, uint64_t* o) { hipLaunchKernelGGL(k_all_same, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d);
        const double f2 = run([](int b, uint64_t* o) { hipLaunchKernelGGL(k_free, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d);
        printf("waves/SIMD %d: free %.3f ms (again %.3f), ab_same %.3f (x%.3f), a_on_acc %.3f (x%.3f), all_same %.3f (x%.3f)\n", wps, f, f2, s, s / f, a, a / f, l, l / f);
    }
    return 0;
}
