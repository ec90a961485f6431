// Host-pointer side of the C ABI: what the reference's Rust host gets when `create_proof` stays on the CPU and only
// `best_multiexp` / `best_fft` / `Params::commit*` / `EvaluationDomain::*` cross into libtrh with slices in HOST memory
// (north_star's integration; the call site is /root/reference/src/test_utils.rs:41-49, the slices are the `Vec<F>`s of
// halo2_proofs 0.2.0 `Polynomial`s).  Every byte crosses PCIe here, so this file is about the link, not about arithmetic:
//
//   * staging (stage_h2d / stage_d2h): a pageable buffer travels through a ring of pinned slots; a small pool of host threads
//     copies slot i + 1 while the DMA engine moves slot i.  Measured on the MI355X box (profiles/pcie_probe_r03.txt): the HIP
//     runtime's own pageable path pins the caller's pages the first time it sees them (27 GB/s up, 12.7 GB/s down into
//     untouched pages; the prover's columns are fresh allocations every time) and hipMemcpyAsync on pageable memory blocks; pinned
//     copies run at 57 GB/s, both directions at once at 2 x 48 GB/s.  Memory the caller pinned itself (trh_host_register /
//     trh_host_alloc) skips the ring.
//   * batch pipelines (host_pipeline): uploads, kernels and downloads of consecutive columns run on three streams over a ring
//     of device buffers; the calling thread uploads and launches, a helper thread drains the downloads -- the link is used in
//     both directions at once and the kernels hide under it.
//   * tiled MSMs (msm_host_tiled): an MSM over host scalars (and host bases) is a sum over ranges; range t + 1 is uploaded
//     while range t is computed.
#include <string.h>

#include <array>
#include <atomic>
#include <chrono>
#include <deque>
#include <condition_variable>
#include <string>
#include <thread>

#include "copypool.h"
#include "ctx.h"

namespace trh {

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int copy_threads() {
    const int v = opt().copy_threads;
    if (v >= 0 && v <= 64) return v;
    const unsigned hw = std::thread::hardware_concurrency();
    // one memcpy thread moves 24-36 GB/s on the box, five (four workers + the caller) 60-85: above the 57 GB/s of the link
    return hw >= 16 ? 4 : hw >= 8 ? 2 : hw >= 4 ? 1 : 0;
}

// TRH_TRACE bit 0: the single-call host entries print their timeline (microseconds since the call began) to stderr
thread_local double t_trace_t0 = 0;   // > 0 while a traced call is running on this thread: the staging loops then report every chunk
struct IoTrace {
    bool on;
    double t0;
    IoTrace() : on((opt().trace & 1) != 0), t0(now_s()) { if (on) t_trace_t0 = t0; }
    ~IoTrace() { t_trace_t0 = 0; }
    void mark(const char* what, size_t bytes = 0) const {
        if (on) fprintf(stderr, "[trh io] %9.1f us  %s %zu\n", (now_s() - t0) * 1e6, what, bytes);
    }
};
inline void trace_chunk(const char* what, size_t k, size_t bytes) {
    if (t_trace_t0 > 0) fprintf(stderr, "[trh io] %9.1f us    %s %zu (%zu bytes)\n", (now_s() - t_trace_t0) * 1e6, what, k, bytes);
}

// streaming stores for the copies out of the download ring (into the caller's memory, bit 0) and into the upload ring (bit 1): both on
// (measured in round 4; the cached forms were the A/B)
constexpr int nt_mode() { return 3; }

// one 64-byte line per 64 KiB, the first and the last: zero?  (a hint, never a proof: stage_h2d's `speculate`)
bool probe_zero(const char* p, size_t bytes) {
    const uint64_t* first = (const uint64_t*)p;
    const uint64_t* last = (const uint64_t*)(p + ((bytes - 64) & ~(size_t)7));
    for (int k = 0; k < 8; ++k) if (first[k] | last[k]) return false;
    for (size_t o = (size_t)64 << 10; o + 64 <= bytes; o += (size_t)64 << 10) {
        const uint64_t* w = (const uint64_t*)(p + o);
        if (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) return false;
    }
    return true;
}

// is this host pointer already page-locked (hipHostMalloc / hipHostRegister)?  Then the DMA engine reads it directly.
bool is_pinned(const void* p) {
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}

}  // namespace

int stage_ensure(Ctx& c) {
    Stage& st = c.stage;
    if (st.up) return TRH_OK;
    if (!st.up_pool) st.up_pool = new CopyPool(copy_threads());
    if (!st.down_pool) st.down_pool = new CopyPool(copy_threads());
    // x NS = 4 slots per direction; smaller slots lose to the per-slot hand-over (measured: 16 MiB 32 ms, 8 MiB 39 ms, 4 MiB 45 ms for the 1.6 GB of a 2^24 best_multiexp)
    size_t slot = (size_t)16 << 20;
    if (opt().stage_slot_mb >= 1 && opt().stage_slot_mb <= 256) slot = (size_t)opt().stage_slot_mb << 20;
    hipError_t e = hipHostMalloc((void**)&st.up, slot * Stage::NS, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&st.down, slot * Stage::NS, hipHostMallocDefault);
    for (int i = 0; i < Stage::NS && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&st.up_ev[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&st.down_ev[i], hipEventDisableTiming);
    }
    for (int i = 0; i < 4 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&st.ev_up[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&st.ev_comp[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st.us, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st.cs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st.ds, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("host staging: %s", hipGetErrorString(e));
        stage_release(c);
        return e == hipErrorOutOfMemory ? TRH_ENOMEM : TRH_EHIP;
    }
    st.slot = slot;
    return TRH_OK;
}

void stage_release(Ctx& c) {
    Stage& st = c.stage;
    delete st.up_pool; delete st.down_pool;  // stops and joins this context's copy threads
    st.up_pool = st.down_pool = nullptr;
    if (st.up) (void)hipHostFree(st.up);
    if (st.down) (void)hipHostFree(st.down);
    st.up = st.down = nullptr;
    for (auto& kv : st.xfer_ev) for (hipEvent_t e : kv.second) if (e) (void)hipEventDestroy(e);
    st.xfer_ev.clear();
    for (int i = 0; i < Stage::NS; ++i) {
        if (st.up_ev[i]) (void)hipEventDestroy(st.up_ev[i]);
        if (st.down_ev[i]) (void)hipEventDestroy(st.down_ev[i]);
        st.up_ev[i] = st.down_ev[i] = nullptr;
        st.up_used[i] = false;
    }
    for (int i = 0; i < 4; ++i) {
        if (st.ev_up[i]) (void)hipEventDestroy(st.ev_up[i]);
        if (st.ev_comp[i]) (void)hipEventDestroy(st.ev_comp[i]);
        st.ev_up[i] = st.ev_comp[i] = nullptr;
        st.ring_in[i].release(); st.ring_out[i].release();
    }
    if (st.us) (void)hipStreamDestroy(st.us);
    if (st.cs) (void)hipStreamDestroy(st.cs);
    if (st.ds) (void)hipStreamDestroy(st.ds);
    st.us = st.cs = st.ds = nullptr;
    st.slot = 0;
}

int stage_h2d(Ctx& c, void* dst_dev, const void* src_host, size_t bytes, hipStream_t s, bool part_of_batch, bool zero_elide, bool head_only,
              std::vector<std::pair<size_t, size_t>>* speculate) {
    if (!bytes) return TRH_OK;
    TRH_TRY(stage_ensure(c));
    Stage& st = c.stage;
    const double t0 = now_s();
    if (is_pinned(src_host)) {
        TRH_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
    } else {
        // a lone transfer starts (and ends) with short chunks: the copy into slot i + 1 hides under the DMA of slot i, and nothing hides the
        // first copy; inside a batch the previous call's DMA is still running, so a transfer that fits one slot is not split
        std::vector<size_t> plan;
        chunk_plan(bytes, st.slot, !part_of_batch, !part_of_batch && !head_only, plan, (zero_elide && !part_of_batch) ? (size_t)256 << 10 : 0);
        size_t off = 0;
        for (const size_t cur : plan) {
            if (speculate && cur >= ((size_t)1 << 20) && probe_zero((const char*)src_host + off, cur)) {
                // looks like padding: cleared on the device NOW, read through LATER (the caller verifies the range while the device works)
                TRH_HIP_TRY(hipMemsetAsync((char*)dst_dev + off, 0, cur, s));
                speculate->push_back({off, cur});
                st.up_zero_bytes += (double)cur;
                off += cur;
                continue;
            }
            const int sl = (int)(st.up_next % Stage::NS);
            if (st.up_used[sl]) TRH_HIP_TRY(hipEventSynchronize(st.up_ev[sl]));
            trace_chunk("up: slot free, copy begins", off, cur);
            char* pin = st.up + (size_t)sl * st.slot;
            if (st.up_pool->copy(pin, (const char*)src_host + off, cur, zero_elide, (nt_mode() & 2) != 0)) {
                // zero throughout (the padding of a zero-padded vector): cleared on the device, nothing crosses the link, the slot stays free
                TRH_HIP_TRY(hipMemsetAsync((char*)dst_dev + off, 0, cur, s));
                st.up_zero_bytes += (double)cur;
            } else {
                TRH_HIP_TRY(hipMemcpyAsync((char*)dst_dev + off, pin, cur, hipMemcpyHostToDevice, s));
                TRH_HIP_TRY(hipEventRecord(st.up_ev[sl], s));
                st.up_used[sl] = true;
                ++st.up_next;
            }
            off += cur;
        }
    }
    st.up_bytes += (double)bytes;
    st.up_s += now_s() - t0;
    return TRH_OK;
}

int stage_d2h(Ctx& c, void* dst_host, const void* src_dev, size_t bytes, hipStream_t s, const std::function<int()>* before_copy_out) {
    if (!bytes) return TRH_OK;
    TRH_TRY(stage_ensure(c));
    Stage& st = c.stage;
    const double t0 = now_s();
    if (is_pinned(dst_host)) {
        if (before_copy_out) { const int rc = (*before_copy_out)(); if (rc != TRH_OK) return rc; }
        TRH_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
        TRH_HIP_TRY(hipStreamSynchronize(s));
    } else {
        std::vector<size_t> plan, offs;
        chunk_plan(bytes, st.slot, false, true, plan);  // short LAST chunks: the caller waits for the copy out of the last one
        size_t o = 0;
        for (const size_t cur : plan) { offs.push_back(o); o += cur; }
        const size_t nchunks = plan.size();
        size_t issued = 0;
        for (size_t k = 0; k < nchunks; ++k) {
            for (; issued < nchunks && issued < k + Stage::NS; ++issued) {  // keep the ring full
                const int sl = (int)(issued % Stage::NS);
                TRH_HIP_TRY(hipMemcpyAsync(st.down + (size_t)sl * st.slot, (const char*)src_dev + offs[issued], plan[issued], hipMemcpyDeviceToHost, s));
                TRH_HIP_TRY(hipEventRecord(st.down_ev[sl], s));
            }
            if (k == 0 && before_copy_out) {  // the first DMAs are on their way into the ring; nothing has been written to dst_host yet
                const int rc = (*before_copy_out)();
                trace_chunk("down: before-copy-out hook returned", (size_t)rc, 0);
                if (rc != TRH_OK) { (void)hipStreamSynchronize(s); return rc; }
            }
            const int sl = (int)(k % Stage::NS);
            TRH_HIP_TRY(hipEventSynchronize(st.down_ev[sl]));
            trace_chunk("down: DMA of chunk done", k, plan[k]);
            st.down_pool->copy((char*)dst_host + offs[k], st.down + (size_t)sl * st.slot, plan[k], false, (nt_mode() & 1) != 0);
            trace_chunk("down: copied out", k, plan[k]);
        }
    }
    st.down_bytes += (double)bytes;
    st.down_s += now_s() - t0;
    return TRH_OK;
}

// Device-to-device hand-over between two GPUs WITHOUT peer access (or with TRH_FORCE_NO_PEER=1): the bytes travel through the slots of the
// destination context's pinned download ring -- D2H on the source device's stream, H2D on the destination's -- chained by events only:
// slot k is written after the H2D that last read it (event wait on the source stream) and read after its D2H (event wait on the
// destination stream).  No host thread waits; both directions of both links run at once.
int stage_d2d_via_host(Ctx& dstc, void* dst_dev, hipStream_t dst_stream, const void* src_dev, int src_device, hipStream_t src_stream, size_t bytes) {
    if (!bytes) return TRH_OK;
    TRH_TRY(stage_ensure(dstc));
    Stage& st = dstc.stage;
    int cur_dev = -1;
    TRH_HIP_TRY(hipGetDevice(&cur_dev));
    // events recorded on the SOURCE device's stream have to belong to that device: one set per source device, kept until stage_release
    auto it = st.xfer_ev.find(src_device);
    if (it == st.xfer_ev.end()) {
        std::array<hipEvent_t, Stage::NS> evs{};
        if (hipSetDevice(src_device) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipSetDevice(cur_dev);  // every exit leaves the caller's device current
            set_error("hand-over through the host: hipSetDevice(%d) failed", src_device);
            return TRH_EHIP;
        }
        for (int i = 0; i < Stage::NS; ++i) {
            const hipError_t e = hipEventCreateWithFlags(&evs[i], hipEventDisableTiming);
            if (e != hipSuccess) {
                for (int j = 0; j < i; ++j) (void)hipEventDestroy(evs[j]);
                (void)hipSetDevice(cur_dev);
                set_error("hand-over through the host: hipEventCreate: %s", hipGetErrorString(e));
                return TRH_EHIP;
            }
        }
        it = st.xfer_ev.emplace(src_device, evs).first;
    }
    hipEvent_t* filled = it->second.data();
    int rc = TRH_OK;
    size_t k = 0;
    for (size_t off = 0; off < bytes && rc == TRH_OK; off += st.slot, ++k) {
        const size_t cur = bytes - off < st.slot ? bytes - off : st.slot;
        const int sl = (int)(k % Stage::NS);
        char* pin = st.down + (size_t)sl * st.slot;
        hipError_t e = hipSetDevice(src_device);
        // the transfer that used this slot last (an earlier chunk's H2D, an earlier hand-over's, a download's DMA); an event never recorded does not wait
        if (e == hipSuccess) e = hipStreamWaitEvent(src_stream, st.down_ev[sl], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(pin, (const char*)src_dev + off, cur, hipMemcpyDeviceToHost, src_stream);
        if (e == hipSuccess) e = hipEventRecord(filled[sl], src_stream);
        if (e == hipSuccess) e = hipSetDevice(dstc.device);
        if (e == hipSuccess) e = hipStreamWaitEvent(dst_stream, filled[sl], 0);
        if (e == hipSuccess) e = hipMemcpyAsync((char*)dst_dev + off, pin, cur, hipMemcpyHostToDevice, dst_stream);
        if (e == hipSuccess) e = hipEventRecord(st.down_ev[sl], dst_stream);
        if (e != hipSuccess) { set_error("hand-over through the host: %s", hipGetErrorString(e)); rc = TRH_EHIP; }
    }
    // the ring's next user runs behind dst_stream: the caller entered the destination context with it, so the context's order event is
    // recorded there when the entry returns, and every later entry (and stage_begin) waits for that event
    (void)hipSetDevice(cur_dev);
    st.up_bytes += (double)bytes;
    st.down_bytes += (double)bytes;
    return rc;
}

// the three streams of the stage join the context's order: whatever ran last on the context's scratch finishes first
int stage_begin(Ctx& c) {
    TRH_TRY(stage_ensure(c));
    Stage& st = c.stage;
    if (c.last_stream_valid) {
        TRH_HIP_TRY(hipStreamWaitEvent(st.us, c.order_ev, 0));
        TRH_HIP_TRY(hipStreamWaitEvent(st.cs, c.order_ev, 0));
        TRH_HIP_TRY(hipStreamWaitEvent(st.ds, c.order_ev, 0));
    }
    return TRH_OK;
}
int stage_end(Ctx& c) {
    Stage& st = c.stage;
    TRH_HIP_TRY(hipStreamSynchronize(st.us));
    TRH_HIP_TRY(hipStreamSynchronize(st.cs));
    TRH_HIP_TRY(hipStreamSynchronize(st.ds));
    return TRH_OK;
}

int host_pipeline(Ctx& c, const HostPipe& p) {
    if (!p.count) return TRH_OK;
    TRH_TRY(stage_begin(c));
    Stage& st = c.stage;
    const size_t D = p.count < 3 ? p.count : 3;
    for (size_t d = 0; d < D; ++d) {
        TRH_TRY(st.ring_in[d].ensure(p.in_bytes));
        if (!p.in_place) TRH_TRY(st.ring_out[d].ensure(p.out_bytes));
    }
    std::mutex mu;
    std::condition_variable cv;
    size_t submitted = 0, downloaded = 0;
    bool abort = false;
    int helper_rc = TRH_OK;
    std::string helper_err;
    const int device = c.device;
    std::thread helper([&] {
        int rc = hipSetDevice(device) == hipSuccess ? TRH_OK : TRH_EHIP;
        struct Chunk { int slot; char* dst; size_t bytes; size_t item; bool direct; bool marker; };
        std::deque<Chunk> inflight;
        std::vector<HostPipe::Seg> segs;
        std::vector<size_t> chunks_left(p.count, 0);
        size_t next_item = 0, seg_idx = 0, seg_off = 0, issue_ctr = 0, items_done = 0;
        bool have_segs = false;
        const double t_begin = now_s();
        double bytes_moved = 0;
        auto fail = [&](int code) {
            std::lock_guard<std::mutex> lk(mu);
            if (helper_rc == TRH_OK) { helper_rc = code; helper_err = trh_last_error(); }
            abort = true;
        };
        if (rc != TRH_OK) { set_error("host pipeline: hipSetDevice failed"); fail(rc); cv.notify_all(); return; }
        while (items_done < p.count) {
            // keep the ring full: chunks of the next items as soon as their kernels are queued
            while ((int)inflight.size() < Stage::NS && (have_segs || next_item < p.count)) {
                if (!have_segs) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        if (inflight.empty()) cv.wait(lk, [&] { return submitted > next_item || abort; });
                        if (abort && submitted <= next_item) return;
                        if (submitted <= next_item) break;  // not launched yet: go drain what is in flight
                    }
                    const size_t slot = next_item % D;
                    if (hipStreamWaitEvent(st.ds, st.ev_comp[slot], 0) != hipSuccess) { set_error("host pipeline: hipStreamWaitEvent failed"); fail(TRH_EHIP); cv.notify_all(); return; }
                    segs.clear();
                    p.segments(next_item, p.in_place ? st.ring_in[slot].p : st.ring_out[slot].p, segs);
                    size_t w = 0;  // zero-byte segments carry nothing: dropped here, so that they neither count as a chunk nor complete one early
                    for (size_t r = 0; r < segs.size(); ++r) if (segs[r].bytes) segs[w++] = segs[r];
                    segs.resize(w);
                    size_t nch = 0;
                    for (const HostPipe::Seg& sg : segs) nch += is_pinned(sg.dst) ? 1 : (sg.bytes + st.slot - 1) / st.slot;
                    chunks_left[next_item] = nch;
                    have_segs = true; seg_idx = 0; seg_off = 0;
                    if (nch == 0) {
                        // an item with nothing to download completes IN ORDER: behind the chunks of the items before it (publishing it at once
                        // would free the ring slot of an earlier item that is still in flight)
                        have_segs = false;
                        chunks_left[next_item] = 1;
                        inflight.push_back(Chunk{0, nullptr, 0, next_item, true, true});
                        ++next_item;
                        continue;
                    }
                }
                const HostPipe::Seg& sg = segs[seg_idx];
                const bool direct = is_pinned(sg.dst);
                const size_t cur = direct ? sg.bytes : (sg.bytes - seg_off < st.slot ? sg.bytes - seg_off : st.slot);
                const int sl = (int)(issue_ctr++ % Stage::NS);
                char* land = direct ? (char*)sg.dst : st.down + (size_t)sl * st.slot;
                hipError_t e = hipMemcpyAsync(land, (const char*)sg.src + seg_off, cur, hipMemcpyDeviceToHost, st.ds);
                if (e == hipSuccess) e = hipEventRecord(st.down_ev[sl], st.ds);
                if (e != hipSuccess) { set_error("host pipeline: download failed: %s", hipGetErrorString(e)); fail(TRH_EHIP); cv.notify_all(); return; }
                inflight.push_back(Chunk{sl, (char*)sg.dst + seg_off, cur, next_item, direct, false});
                seg_off += cur;
                if (seg_off >= sg.bytes) { seg_off = 0; if (++seg_idx >= segs.size()) { have_segs = false; ++next_item; } }
            }
            if (inflight.empty()) continue;
            const Chunk ch = inflight.front();
            if (!ch.marker && hipEventSynchronize(st.down_ev[ch.slot]) != hipSuccess) { set_error("host pipeline: hipEventSynchronize failed"); fail(TRH_EHIP); cv.notify_all(); return; }
            if (!ch.direct) st.down_pool->copy(ch.dst, st.down + (size_t)ch.slot * st.slot, ch.bytes, false, (nt_mode() & 1) != 0);
            bytes_moved += (double)ch.bytes;
            inflight.pop_front();
            if (--chunks_left[ch.item] == 0) {
                ++items_done;
                { std::lock_guard<std::mutex> lk(mu); downloaded = ch.item + 1; }
                cv.notify_all();
            }
        }
        st.down_bytes += bytes_moved;
        st.down_s += now_s() - t_begin;
    });
    int rc = TRH_OK;
    auto step = [&](size_t i) -> int {
        const size_t slot = i % D;
        if (i >= D) {  // the slot's buffers are free once item i - D is back on the host
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return downloaded + D > i || abort; });
            if (abort) return helper_rc;
        }
        TRH_TRY(p.upload(i, st.ring_in[slot].p));
        TRH_HIP_TRY(hipEventRecord(st.ev_up[slot], st.us));
        TRH_HIP_TRY(hipStreamWaitEvent(st.cs, st.ev_up[slot], 0));
        TRH_TRY(p.compute(i, st.ring_in[slot].p, p.in_place ? st.ring_in[slot].p : st.ring_out[slot].p, st.cs));
        TRH_HIP_TRY(hipEventRecord(st.ev_comp[slot], st.cs));
        {
            std::lock_guard<std::mutex> lk(mu);
            submitted = i + 1;
        }
        cv.notify_all();
        return TRH_OK;
    };
    for (size_t i = 0; i < p.count && rc == TRH_OK; ++i) rc = step(i);
    if (rc != TRH_OK) {
        std::lock_guard<std::mutex> lk(mu);
        abort = true;
    }
    cv.notify_all();
    helper.join();
    if (rc == TRH_OK && helper_rc != TRH_OK) { set_error("%s", helper_err.c_str()); rc = helper_rc; }
    const int rc2 = stage_end(c);
    return rc != TRH_OK ? rc : rc2;
}

// halo2_proofs::arithmetic::best_fft on a host slice: up, transform, down.  One transform cannot overlap its own transfers (its
// first pass reads elements from the whole array, its last pass writes the whole array), so this is the sum of the three; the
// batch form below overlaps them across columns.
int best_fft_host(int field, uint64_t* a, const uint64_t* omega, uint32_t log_n) {
    Range range(field == TRH_FP ? "trh_best_fft_fp" : "trh_best_fft_fq");
    if (!a || !omega) { set_error("best_fft: null pointer"); return TRH_EINVAL; }
    if (log_n > 27) { set_error("best_fft: log_n %u > 27 unsupported", log_n); return TRH_EINVAL; }
    TRH_ENTER(0);
    Ctx& c = ctx();
    TRH_TRY(stage_begin(c));
    StageScope scope(c);
    hipStream_t s = c.stage.cs;
    const size_t bytes = (size_t)32 << log_n;
    IoTrace tr;
    TRH_TRY(c.io.ensure(bytes));
    tr.mark("begin, bytes", bytes);
    // Zero slots are not sent (coeff_to_extended hands over a vector that is zero beyond its first 2^k entries: 7/8 of the upload).  Reading
    // 56 MiB of zeros to be SURE they are zeros takes the host 0.5 ms, and nothing else could start before it: so chunks that look like
    // padding (a sparse probe) are cleared on the device at once, the transform and the first downloads are queued, and the padding is
    // read through while the device works -- before anything is written back into `a` (the transform is in place: until then `a` still
    // holds the input).  If a probed chunk turns out not to be zero the speculative result is dropped and the call starts over plainly.
    std::vector<std::pair<size_t, size_t>> spec;
    const bool speculative = bytes >= ((size_t)8 << 20) && !is_pinned(a);
    const double up_bytes0 = c.stage.up_bytes, up_zero0 = c.stage.up_zero_bytes;
    TRH_TRY(stage_h2d(c, c.io.p, a, bytes, s, false, true, false, speculative ? &spec : nullptr));
    tr.mark("upload issued; speculated ranges", spec.size());
    TRH_TRY(ntt_device(field, c.io.p, log_n, omega, 1, s));
    tr.mark("transform queued");
    // the hook's own verdict travels in a flag, not in a public error code: a TRH_EBUSY out of stage_d2h stays what it says
    bool guess_wrong = false;
    const std::function<int()> verify = [&]() -> int {
        for (const auto& r : spec)
            if (!c.stage.up_pool->copy(nullptr, (const char*)a + r.first, r.second, true, false, true)) { guess_wrong = true; return TRH_EINVAL; }
        return TRH_OK;
    };
    int rc = stage_d2h(c, a, c.io.p, bytes, s, spec.empty() ? nullptr : &verify);
    if (guess_wrong) {  // a chunk that probed as zero was not: `a` is untouched, do it again without guessing
        tr.mark("speculation failed: plain pass");
        TRH_HIP_TRY(hipStreamSynchronize(s));
        // trh_io_stats counts the call's bytes once: the dropped first pass leaves the counters (its host seconds stay -- they were spent)
        c.stage.up_bytes = up_bytes0;
        c.stage.up_zero_bytes = up_zero0;
        TRH_TRY(stage_h2d(c, c.io.p, a, bytes, s, false, true));
        TRH_TRY(ntt_device(field, c.io.p, log_n, omega, 1, s));
        rc = stage_d2h(c, a, c.io.p, bytes, s);
    }
    TRH_TRY(rc);
    tr.mark("download complete");
    const int rc_end = scope.finish();
    tr.mark("end");
    return rc_end;
}

static int best_fft_batch_host(int field, uint64_t* const* a, size_t count, const uint64_t* omega, uint32_t log_n) {
    Range range("trh_best_fft_batch");
    if (!omega || (count && !a)) { set_error("best_fft_batch: null pointer"); return TRH_EINVAL; }
    if (log_n > 27) { set_error("best_fft_batch: log_n %u > 27 unsupported", log_n); return TRH_EINVAL; }
    for (size_t i = 0; i < count; ++i) if (!a[i]) { set_error("best_fft_batch: column %zu is null", i); return TRH_EINVAL; }
    TRH_ENTER(0);
    Ctx& c = ctx();
    const size_t bytes = (size_t)32 << log_n;
    // small transforms travel in groups, so that a pipeline item is worth a few MiB of link time
    size_t group = 1;
    while (group < 64 && group * bytes < ((size_t)32 << 20)) group <<= 1;
    HostPipe p;
    p.count = (count + group - 1) / group;
    p.in_bytes = group * bytes;
    p.in_place = true;
    p.upload = [&](size_t it, void* din) -> int {
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) TRH_TRY(stage_h2d(c, (char*)din + (j - it * group) * bytes, a[j], bytes, c.stage.us, true));
        return TRH_OK;
    };
    p.compute = [&](size_t it, void* din, void*, hipStream_t s) -> int {
        const size_t nb = count - it * group < group ? count - it * group : group;
        return ntt_device(field, din, log_n, omega, nb, s);
    };
    p.segments = [&](size_t it, const void* dout, std::vector<HostPipe::Seg>& out) {
        for (size_t j = it * group; j < count && j < (it + 1) * group; ++j) out.push_back(HostPipe::Seg{a[j], (const char*)dout + (j - it * group) * bytes, bytes});
    };
    return host_pipeline(c, p);
}

}  // namespace trh

using namespace trh;

extern "C" {

int trh_best_fft_batch_fp(uint64_t* const* a, size_t count, const uint64_t omega[4], uint32_t log_n) { return best_fft_batch_host(TRH_FP, a, count, omega, log_n); }
int trh_best_fft_batch_fq(uint64_t* const* a, size_t count, const uint64_t omega[4], uint32_t log_n) { return best_fft_batch_host(TRH_FQ, a, count, omega, log_n); }

int trh_host_register(void* host, size_t bytes) {
    TRH_TRY(require_init());
    if (!host || !bytes) { set_error("host_register: bad arguments"); return TRH_EINVAL; }
    TRH_ENTER(0);
    TRH_HIP_TRY(hipHostRegister(host, bytes, hipHostRegisterDefault));
    return TRH_OK;
}
int trh_host_unregister(void* host) {
    TRH_TRY(require_init());
    TRH_ENTER(0);
    TRH_HIP_TRY(hipHostUnregister(host));
    return TRH_OK;
}
int trh_host_alloc(void** host, size_t bytes) {
    TRH_TRY(require_init());
    if (!host) { set_error("host_alloc: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(0);
    const hipError_t e = hipHostMalloc(host, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { set_error("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return TRH_ENOMEM; }
    return TRH_OK;
}
int trh_host_free(void* host) {
    TRH_TRY(require_init());
    TRH_ENTER(0);
    TRH_HIP_TRY(hipHostFree(host));
    return TRH_OK;
}

int trh_io_stats(trh_io_stats_t* out, int reset) {
    if (!out) { set_error("io_stats: null pointer"); return TRH_EINVAL; }
    TRH_ENTER(0);
    Stage& st = ctx().stage;
    out->h2d_bytes = st.up_bytes; out->d2h_bytes = st.down_bytes; out->h2d_seconds = st.up_s; out->d2h_seconds = st.down_s;
    out->h2d_zero_bytes = st.up_zero_bytes;
    if (reset) st.up_bytes = st.down_bytes = st.up_s = st.down_s = st.up_zero_bytes = 0;
    return TRH_OK;
}

}  // extern "C"
