// What does a 32-bit ALU instruction cost NEXT TO 64-bit multiply-adds? (bench.py priced the non-mad VALU
// instructions of msm_accumulate_seg_kernel at the 65 T/s of a pure v_add_u32 loop, DESIGN section 6 at one 4-cycle issue slot each.)
// Pure-register loops at 2 / 3 / 4 / 8 waves per SIMD:
//   mad        8 independent v_mad_i64_i32 per iteration
//   add        8 independent v_add_u32
//   mix_indep  8 mads + 4 adds, all independent                       (the kernel's 1170 : 563 ratio)
//   mix_dep    the reduction-round shape: and -> shift-add -> 5 mads that use the and's result, round after round
//   mix_<op>   8 mads + 4 <op> for the other instruction classes of the kernel's carry handling: v_and_b32, v_ashrrev_i64 (64-bit shift),
//              v_lshl_add_u64 (64-bit add), v_mov_b32
// Reports, per row, the AVERAGE SIMD cycles per wave instruction and -- what the issue model needs (round 3 read the
// average of the 8 + 4 mix as the cost of the add) -- the MARGINAL cost of the non-mad instruction:
//     (cycles of the mixed loop - mads x cycles per mad in the pure-mad loop at the same occupancy) / others
// once from clock64 (wave lifetimes, independent of the clock the run happened to get) and once from the wall times of the launches.
// build: hipcc -O3 --offload-arch=gfx950 tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int64_t i64;
typedef int32_t i32;
typedef uint32_t u32;
typedef uint64_t u64;
constexpr int ITERS = 32768;
__device__ unsigned long long g_cycles[4096 * 4];

__device__ __forceinline__ void mac(i64& acc, i32 a, i32 b) { asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void add(i32& x, i32 b) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)); }
struct OpAnd { typedef i32 T; static __device__ __forceinline__ void op(T& y, i32 b) { asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(y)); } };
struct OpMov { typedef i32 T; static __device__ __forceinline__ void op(T& y, i32 b) { asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(b)); } };
struct OpShr64 { typedef i64 T; static __device__ __forceinline__ void op(T& y, i32 b) { asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(y)); } };
struct OpAdd64 { typedef i64 T; static __device__ __forceinline__ void op(T& y, i32 b) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(y) : "v"((i64)b)); } };

// 8 mads + 4 Op per iteration, all independent (the shape of k_mix_indep with another instruction class in the add's place)
template <class Op>
__global__ void __launch_bounds__(256) k_mix_op(u64* out, i32 a, i32 b) {
    const u64 t_begin = clock64();
    i64 x[8];
    typename Op::T y[4];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    for (int k = 0; k < 4; ++k) y[k] = threadIdx.x + k;
    i32 aa = a + threadIdx.x, bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { mac(x[2 * k], aa, bb); Op::op(y[k], bb); mac(x[2 * k + 1], aa, bb); }
    }
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r ^= (u64)x[k];
    for (int k = 0; k < 4; ++k) r ^= (u64)y[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) g_cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = clock64() - t_begin;
}

__global__ void __launch_bounds__(256) k_mad(u64* out, i32 a, i32 b) {
    const u64 t_begin = clock64();
    i64 x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    i32 aa = a + threadIdx.x, bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) mac(x[k], aa, bb);
    }
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r ^= (u64)x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) g_cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = clock64() - t_begin;
}
__global__ void __launch_bounds__(256) k_add(u64* out, i32 a, i32 b) {
    const u64 t_begin = clock64();
    i32 y[8];
    for (int k = 0; k < 8; ++k) y[k] = threadIdx.x + k;
    i32 bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) add(y[k], bb);
    }
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r ^= (u64)(u32)y[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) g_cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = clock64() - t_begin;
}
__global__ void __launch_bounds__(256) k_mix_indep(u64* out, i32 a, i32 b) {
    const u64 t_begin = clock64();
    i64 x[8];
    i32 y[4];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    for (int k = 0; k < 4; ++k) y[k] = threadIdx.x + k;
    i32 aa = a + threadIdx.x, bb = b + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { mac(x[2 * k], aa, bb); add(y[k], bb); mac(x[2 * k + 1], aa, bb); }
    }
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r ^= (u64)x[k];
    for (int k = 0; k < 4; ++k) r ^= (u64)(u32)y[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) g_cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = clock64() - t_begin;
}
// the shape of fy_reduce_sub's rounds: r = col & MASK; next += col >> 29; five multiply-adds with r
__global__ void __launch_bounds__(256) k_mix_dep(u64* out, i32 a, i32 b) {
    const u64 t_begin = clock64();
    i64 c[10];
    for (int k = 0; k < 10; ++k) c[k] = ((i64)threadIdx.x << 33) + k;
    i32 p1 = a + 1, p2 = a + 2, p3 = a + 3, p4 = a + 4, p8 = b;
    for (int i = 0; i < ITERS / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // 8 rounds over a rotating window of columns
            i64& col = c[j % 10];
            i32 r;
            asm volatile("v_and_b32 %0, 0x1fffffff, %1" : "=v"(r) : "v"((i32)col));
            i64 sh;
            asm volatile("v_ashrrev_i64 %0, 29, %1" : "=v"(sh) : "v"(col));
            asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(c[(j + 1) % 10]) : "v"(sh));
            mac(c[(j + 1) % 10], r, p1); mac(c[(j + 2) % 10], r, p2); mac(c[(j + 3) % 10], r, p3); mac(c[(j + 4) % 10], r, p4); mac(c[(j + 8) % 10], r, p8);
        }
    }
    u64 r = 0;
    for (int k = 0; k < 10; ++k) r ^= (u64)c[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) g_cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = clock64() - t_begin;
}
__global__ void k_clock(u64* out) {
    const u64 t0 = wall_clock64();
    const u64 c0 = clock64();
    u32 x = threadIdx.x;
    for (int i = 0; i < 200000; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
    const u64 c1 = clock64();
    const u64 t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; out[2] = x; }
}

template <class K>
double run(K kern, int blocks, u64* d_out, double* cycles) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern(blocks, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) kern(blocks, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long h[4096 * 4];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cycles), sizeof(u64) * blocks * 4));
    double sum = 0;
    for (int i = 0; i < blocks * 4; ++i) sum += (double)h[i];
    *cycles = sum / (blocks * 4);  // average lifetime of a wave in shader cycles (all waves of a SIMD are resident at once)
    return ms / 3;
}

int main() {
    u64* d_out;
    CK(hipMalloc(&d_out, sizeof(u64) * 4096 * 1024));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    printf("device %s, %d CUs, clockRate %d kHz, wall clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, wall_khz);
    printf("avg = SIMD cycles per wave instruction over ALL instructions of the loop; marginal = cycles the non-mad instruction adds to a stream of mads\n"
           "  = (loop cycles - mads x cycles per mad of the pure-mad row at the same occupancy) / others; `norm` rescales so that the mad row is 4.00\n");
    const int cus = prop.multiProcessorCount;
    constexpr int NR = 8;
    for (int wps : {2, 3, 4, 8}) {
        const int blocks = cus * wps;  // 256-thread workgroups = 4 waves = one per SIMD
        struct { const char* name; double mads, others; double ms; } rows[NR] = {
            {"mad", 8.0 * ITERS, 0, 0}, {"add", 0, 8.0 * ITERS, 0}, {"mix_indep 8 mad + 4 v_add_u32", 8.0 * ITERS, 4.0 * ITERS, 0}, {"mix_dep 5 mad + 3 alu per round", 5.0 * ITERS, 3.0 * ITERS, 0},
            {"mix 8 mad + 4 v_and_b32", 8.0 * ITERS, 4.0 * ITERS, 0}, {"mix 8 mad + 4 v_mov_b32", 8.0 * ITERS, 4.0 * ITERS, 0},
            {"mix 8 mad + 4 v_ashrrev_i64", 8.0 * ITERS, 4.0 * ITERS, 0}, {"mix 8 mad + 4 v_lshl_add_u64", 8.0 * ITERS, 4.0 * ITERS, 0}};
        double cyc[NR];
        rows[0].ms = run([](int b, u64* o) { hipLaunchKernelGGL(k_mad, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d_out, &cyc[0]);
        rows[1].ms = run([](int b, u64* o) { hipLaunchKernelGGL(k_add, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d_out, &cyc[1]);
        rows[2].ms = run([](int b, u64* o) { hipLaunchKernelGGL(k_mix_indep, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d_out, &cyc[2]);
        rows[3].ms = run([](int b, u64* o) { hipLaunchKernelGGL(k_mix_dep, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d_out, &cyc[3]);
        rows[4].ms = run([](int b, u64* o) { hipLaunchKernelGGL(k_mix_op<OpAnd>, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d_out, &cyc[4]);
        rows[5].ms = run([](int b, u64* o) { hipLaunchKernelGGL(k_mix_op<OpMov>, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d_out, &cyc[5]);
        rows[6].ms = run([](int b, u64* o) { hipLaunchKernelGGL(k_mix_op<OpShr64>, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d_out, &cyc[6]);
        rows[7].ms = run([](int b, u64* o) { hipLaunchKernelGGL(k_mix_op<OpAdd64>, dim3(b), dim3(256), 0, 0, o, 3, 5); }, blocks, d_out, &cyc[7]);
        // every SIMD holds wps waves for the whole kernel: SIMD cycles per wave instruction = wave lifetime / (instructions per wave x wps)
        const double mad_cyc = cyc[0] / (rows[0].mads * wps);      // cycles per mad, pure-mad loop
        const double mad_ms = rows[0].ms / (rows[0].mads * wps);   // the same in wall time
        for (int i = 0; i < NR; ++i) {
            auto& r = rows[i];
            const double per_inst = cyc[i] / ((r.mads + r.others) * wps);
            printf("waves/SIMD %d  %-32s %8.3f ms  avg %5.2f (norm %5.2f) cycles/inst, clock %.2f GHz", wps, r.name, r.ms, per_inst, per_inst * 4.0 / mad_cyc, cyc[i] / (r.ms * 1e-3) / 1e9);
            if (r.others > 0) {
                const double marg_cyc = (cyc[i] / wps - r.mads * mad_cyc) / r.others;
                const double marg_ms = (r.ms / wps - r.mads * mad_ms) / r.others;
                printf("   MARGINAL non-mad: %5.2f cycles (norm %5.2f; from wall times, in units of a mad's time / 4: %5.2f)", marg_cyc, marg_cyc * 4.0 / mad_cyc, marg_ms * 4.0 / mad_ms);
            }
            printf("\n");
        }
    }
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_out);
    CK(hipDeviceSynchronize());
    u64 h[3];
    CK(hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost));
    printf("idle-chip clock: %llu shader cycles in %llu wall ticks\n", (unsigned long long)h[0], (unsigned long long)h[1]);
    return 0;
}
