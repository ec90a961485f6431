// CPU unit test of csrc/hostcombine.h (the 4 x 64-bit host Horner + normalisation) against the generic nine-limb
// implementation of curve.h that the device code shares: random window sums in non-trivial XYZZ form, every window
// width the MSM uses, identity / doubling / cancellation cases.  Built and run by tests/test_hostcombine.py (g++ only).
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../tiny-ram-halo2_amd/csrc/curve.h"
#include "../../tiny-ram-halo2_amd/csrc/hostcombine.h"
using namespace trh;

template <class BF>
static void reference(const XYZZMem* ws, int W, int cb, u64* out_xyz) {  // the former combine_windows_host
    XYZZ<BF> acc = xyzz_identity<BF>();
    for (int j = W - 1; j >= 0; --j) {
        for (int k = 0; k < cb; ++k) acc = xyzz_dbl(acc);
        acc = xyzz_add(acc, xyzz_load<BF>(ws[j]));
    }
    JacobianMem r;
    jac_store(jac_from_affine(xyzz_to_affine(acc)), r);
    memcpy(out_xyz, &r, 96);
}

template <class BF>
static int run(const char* name) {
    int bad = 0;
    Affine<BF> G; G.x = fe_neg(fe_one<BF>()); G.y = fe_dbl(fe_one<BF>());
    u64 seed = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    for (int trial = 0; trial < 40; ++trial) {
        const int cbs[] = {2, 4, 8, 10, 13, 15, 16, 17, 18};
        const int cb = cbs[trial % 9], W = 255 / cb + 1;
        XYZZMem ws[128];
        XYZZ<BF> p = xyzz_from_affine(G);
        for (int j = 0; j < W; ++j) {
            const int steps = 1 + (int)(next() % 5);
            for (int s = 0; s < steps; ++s) { p = xyzz_dbl(p); if (next() & 1) xyzz_madd(p, G); }
            XYZZ<BF> v = p;
            const u64 kind = next() % 8;
            if (kind == 0) v = xyzz_identity<BF>();                 // empty window
            if (kind == 1 && j > 0) v = xyzz_load<BF>(ws[j - 1]);    // repeated value
            xyzz_store(v, ws[j]);
        }
        if (trial == 7) for (int j = 0; j < W; ++j) xyzz_store(xyzz_identity<BF>(), ws[j]);  // all empty -> identity
        if (trial == 8) {  // S_1 = -2^cb S_0-style cancellation on the way: acc + S = identity
            XYZZ<BF> top = xyzz_load<BF>(ws[W - 1]);
            XYZZ<BF> t = top;
            for (int k = 0; k < cb; ++k) t = xyzz_dbl(t);
            xyzz_store(xyzz_neg(t), ws[W - 2]);
        }
        if (trial == 9) {  // acc == S: the addition is a doubling
            XYZZ<BF> t = xyzz_load<BF>(ws[W - 1]);
            for (int k = 0; k < cb; ++k) t = xyzz_dbl(t);
            xyzz_store(t, ws[W - 2]);
        }
        u64 want[12], got[12];
        reference<BF>(ws, W, cb, want);
        hostcombine::combine_windows<BF>((const uint64_t*)ws, W, cb, (uint64_t*)got);
        if (memcmp(want, got, 96) != 0) { ++bad; std::printf("%s: mismatch at trial %d (c = %d)\n", name, trial, cb); }
    }
    // the batch form (one inversion for all items): W = 1 (fixed-base mode) and W = 16, batches around the chunk boundary, identities sprinkled in
    for (int W : {1, 16}) {
        for (size_t batch : {(size_t)1, (size_t)2, (size_t)3, (size_t)64, (size_t)255, (size_t)256, (size_t)257, (size_t)600}) {
            const int cb = 16;
            std::vector<XYZZMem> ws(batch * W);
            XYZZ<BF> p = xyzz_from_affine(G);
            for (size_t i = 0; i < batch * W; ++i) {
                p = xyzz_dbl(p); if (next() & 1) xyzz_madd(p, G);
                XYZZ<BF> v = p;
                if (next() % 7 == 0) v = xyzz_identity<BF>();
                xyzz_store(v, ws[i]);
            }
            if (batch >= 3) for (int j = 0; j < W; ++j) xyzz_store(xyzz_identity<BF>(), ws[(batch - 2) * W + j]);  // one item is the identity altogether
            std::vector<u64> got(batch * 12), want(batch * 12);
            for (size_t i = 0; i < batch; ++i) reference<BF>(ws.data() + i * W, W, cb, want.data() + 12 * i);
            hostcombine::combine_windows_batch<BF>((const uint64_t*)ws.data(), W, cb, batch, (uint64_t*)got.data());
            if (memcmp(want.data(), got.data(), batch * 96) != 0) { ++bad; std::printf("%s: batch mismatch (W = %d, batch = %zu)\n", name, W, batch); }
        }
    }
    return bad;
}

int main() {
    const int bad = run<FpParams>("fp") + run<FqParams>("fq");
    std::printf("hostcombine: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
