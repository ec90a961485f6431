// what lane i of a quad reads under __builtin_amdgcn_update_dpp(quad_perm [P0, P1, P2, P3]) -- the exchanges of csrc/curve_q4.h
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int P0, int P1, int P2, int P3> __device__ int perm(int v) { return __builtin_amdgcn_update_dpp(0, v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[0 * 64 + l] = perm<2, 3, 2, 3>(l);
    out[1 * 64 + l] = perm<0, 1, 0, 0>(l);
    out[2 * 64 + l] = perm<0, 0, 2, 3>(l);
    out[3 * 64 + l] = perm<1, 1, 1, 1>(l);
    out[4 * 64 + l] = perm<0, 0, 0, 0>(l);
    out[5 * 64 + l] = perm<1, 1, 2, 3>(l);
}
int main() {
    int* d; hipMalloc(&d, 6 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"2323", "0100", "0023", "1111", "0000", "1123"};
    for (int p = 0; p < 6; ++p) { printf("%s:", names[p]); for (int l = 0; l < 8; ++l) printf(" %d", h[p * 64 + l]); printf("\n"); }
    return 0;
}
