// The block index behind trh_malloc / trh_free (csrc/devpool.h) with two made-up device ids: eviction takes the freeing device's
// largest idle block first and touches another device's blocks only when it has none (VERDICT r03 "host side" item 9), classes are
// kept apart per device, a block is either live or idle.  Plain C++, run by tests/test_hostcombine.py.
#include <cstdio>

#include "../../tiny-ram-halo2_amd/csrc/devpool.h"

int main() {
    using trh::DevPoolIndex;
    int bad = 0;
    auto expect = [&](bool ok, const char* what) { if (!ok) { std::printf("devpool: FAILED %s\n", what); ++bad; } };
    DevPoolIndex ix;
    char blk[16];
    expect(DevPoolIndex::round(1) == 4096 && DevPoolIndex::round(4097) == 8192 && DevPoolIndex::round((1u << 20) + 1) == (2u << 20), "rounding");
    // device 0: 4 KiB, 1 MiB, 8 MiB; device 1: 4 KiB, 64 MiB
    ix.put_idle(&blk[0], {0, 4096}); ix.put_idle(&blk[1], {0, 1u << 20}); ix.put_idle(&blk[2], {0, 8u << 20});
    ix.put_idle(&blk[3], {1, 4096}); ix.put_idle(&blk[4], {1, 64u << 20});
    expect(ix.idle_bytes == 4096 + (1u << 20) + (8u << 20) + 4096 + (64u << 20), "idle bytes");
    expect(ix.victim(0)->second == &blk[2], "device 0 gives up ITS largest block (8 MiB), not device 1's 64 MiB");
    expect(ix.victim(1)->second == &blk[4], "device 1 gives up its largest block");
    ix.drop(ix.victim(0));
    expect(ix.victim(0)->second == &blk[1], "then its next largest");
    ix.drop(ix.victim(0)); ix.drop(ix.victim(0));
    expect(ix.victim(0)->second == &blk[4], "a device with nothing idle takes the largest block of the highest device");
    expect(ix.victim(2)->second == &blk[4] && ix.victim(-1)->second == &blk[4], "unknown devices likewise");
    // classes are per device
    expect(ix.take(0, 4096) == nullptr, "device 0 has no 4 KiB block left");
    void* p = ix.take(1, 4096);
    expect(p == &blk[3] && ix.live.count(p) == 1 && !ix.is_idle(p), "device 1's 4 KiB block becomes live");
    expect(ix.is_idle(&blk[4]) && !ix.is_idle(&blk[0]), "is_idle");
    expect(ix.idle_bytes == (64u << 20), "idle bytes after the moves");
    std::printf(bad ? "devpool: FAILED (%d)\n" : "devpool: ok\n", bad);
    return bad ? 1 : 0;
}
