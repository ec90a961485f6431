// Bookkeeping of the device blocks trh_malloc / trh_free keep for reuse (capi.hip).  No HIP in here: the index is compiled and run on
// its own by tests/native/devpool_test.cpp with made-up device ids (the GPU box has one device; the eviction order across devices is
// what went wrong in round 3: the "largest block of this device" was the largest block of the highest-numbered device).
#pragma once
#include <stddef.h>

#include <map>
#include <utility>

namespace trh {

struct DevPoolIndex {
    typedef std::pair<int, size_t> Key;                 // (device, rounded size)
    std::map<void*, Key> live;                          // blocks handed out
    std::multimap<Key, void*> idle;                     // blocks waiting for reuse
    size_t idle_bytes = 0;

    static size_t round(size_t bytes) {
        if (bytes < 16) bytes = 16;
        const size_t q = bytes < ((size_t)1 << 20) ? (size_t)4096 : (size_t)1 << 20;
        return (bytes + q - 1) / q * q;
    }
    // an idle block of exactly this class, or null; the block becomes live
    void* take(int device, size_t rounded) {
        auto it = idle.find(Key(device, rounded));
        if (it == idle.end()) return nullptr;
        void* p = it->second;
        idle.erase(it);
        idle_bytes -= rounded;
        live[p] = Key(device, rounded);
        return p;
    }
    void add_live(void* p, int device, size_t rounded) { live[p] = Key(device, rounded); }
    bool is_idle(const void* p) const {
        for (const auto& kv : idle) if (kv.second == p) return true;
        return false;
    }
    // the block to give up when `device` needs room: that device's LARGEST idle block (keys sort by (device, size), so it is the last
    // entry below (device + 1, 0)); a block of another device only when the device has none left.  idle must not be empty.
    std::multimap<Key, void*>::iterator victim(int device) {
        auto it = idle.lower_bound(Key(device + 1, (size_t)0));
        if (it != idle.begin() && std::prev(it)->first.first == device) return std::prev(it);
        return std::prev(idle.end());
    }
    void put_idle(void* p, Key k) { idle.insert({k, p}); idle_bytes += k.second; }
    void drop(std::multimap<Key, void*>::iterator it) { idle_bytes -= it->first.second; idle.erase(it); }
};

}  // namespace trh
