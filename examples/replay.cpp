// Native driver over include/trh.hpp: the arithmetic schedule `halo2_proofs::plonk::create_proof` issues for the
// reference's TinyRamCircuit<WORD_BITS, 8> (reference call site
// /root/reference/src/test_utils.rs:41-49, k = 2 + WORD_BITS / 2 from :20), from a compiled host with no Python in the
// process -- what the Rust prover's side of the boundary looks like.  Synthetic column data; the same primitive kinds,
// sizes and counts as tiny-ram-halo2_amd/replay.py, each kind self-checked once with the host arithmetic of trh.hpp
// or against a second libtrh path (the bit-exact parity against the oracle lives in tests/).
//
//   ./examples/replay [--word-bits 16|32] [--batch 64] [--columns random|witness] [--mode resident|dropin|dropin-batched] [--max-columns N]
//   -> one JSON line, exit code 0 iff every check passed.  `--mode dropin*`: the polynomials stay in host memory and cross PCIe inside
//   the host-pointer entries (run_dropin below)
// `--columns witness`: the value classes the reference's tables hold (flags / WORD_BITS-bit words on the n / 4 live rows, zero padding,
// blinding rows; /root/reference/src/circuits/tables/exe.rs:538-741, tables/prog.rs:139-161) instead of uniformly random columns.
// Besides the per-proof schedule the driver replays keygen_vk / keygen_pk (fixed and sigma columns, the l0 / l_blind / l_last cosets:
// /root/reference/src/test_utils.rs:23-25) and the polynomial side of poly::multiopen::create_proof (x1 fold per point set, kate
// divisions, x2 fold, q' commitment, evaluations at x3, x4 fold) in front of the IPA opening.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "trh.hpp"

using namespace trh;

namespace {

constexpr int N_INSTANCE = 94, N_ADVICE = 263, N_LOOKUPS = 31, N_PERM_PRODUCTS = 47, N_H_PIECES = 5, QUOTIENT_J = 6, N_SYNTH_GATES = 300;
constexpr int N_FIXED = 25, N_SIGMA = 188, BLINDING_ROWS = 6;  // see tiny-ram-halo2_amd/replay.py

// value classes of the 497 Lagrange-basis columns in commitment order (replay.py column_classes)
enum class Kind { Flag, Word, Even, Sorted, Full };
struct ColumnClass { int count; Kind kind; bool blinded; };
const ColumnClass WITNESS_CLASSES[] = {{70, Kind::Flag, false}, {24, Kind::Word, false}, {150, Kind::Flag, true}, {90, Kind::Word, true}, {23, Kind::Even, true},
                                       {58, Kind::Sorted, true}, {4, Kind::Full, true}, {N_LOOKUPS + N_PERM_PRODUCTS, Kind::Full, true}};

struct SplitMix {
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
    Limbs element() { Limbs v{next(), next(), next(), next() >> 2}; return v; }  // < 2^254 < m: a valid residue
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// One step's time on the DEVICE: an event recorded on the (null) stream before the step and one after it, read once the second has
// happened -- what replay.py's torch events measure.  (Until round 6 this was the host's clock between two stream synchronisations: every
// step then started on an idle, down-clocked GPU and paid its own launch latency, 0.1 - 0.4 ms per step and ~5 ms per proof more than the
// Python mirror reported for the same kernels: profiles/r06_driver_diff.md.  --host-clock keeps that clock for the comparison.)
bool g_host_clock = false;
struct Timer {
    double t0 = 0;
    void *e0 = nullptr, *e1 = nullptr;
    Timer() {
        if (g_host_clock) { check(trh_stream_synchronize(nullptr), "sync"); t0 = now_ms(); return; }
        check(trh_event_create(&e0), "event_create"); check(trh_event_create(&e1), "event_create");
        check(trh_event_record(e0, nullptr), "event_record");
    }
    Timer(const Timer&) = delete;
    Timer& operator=(const Timer&) = delete;
    ~Timer() { trh_event_destroy(e0); trh_event_destroy(e1); }
    double stop() {
        if (g_host_clock) { check(trh_stream_synchronize(nullptr), "sync"); return now_ms() - t0; }
        float ms = 0;
        check(trh_event_record(e1, nullptr), "event_record");
        check(trh_event_elapsed_ms(e0, e1, &ms), "event_elapsed_ms");
        return (double)ms;
    }
};

int failures = 0;
void expect(bool ok, const char* what) { if (!ok) { ++failures; std::fprintf(stderr, "CHECK FAILED: %s\n", what); } }

// a stand-in for the BLAKE2b transcript: challenges from a counter
struct Transcript { SplitMix rng{0x7e57}; int points = 0, scalars = 0; };
void tr_write_point(void* c, const uint64_t*) { ++((Transcript*)c)->points; }
void tr_write_scalar(void* c, const uint64_t*) { ++((Transcript*)c)->scalars; }
void tr_squeeze(void* c, uint64_t* out) { const Limbs v = ((Transcript*)c)->rng.element(); std::memcpy(out, v.data(), 32); }
void rng_scalar(void* c, uint64_t* out) { const Limbs v = ((SplitMix*)c)->element(); std::memcpy(out, v.data(), 32); }

// one column of a class as CANONICAL limbs (small integers; `to_montgomery` on the device makes them field elements): live rows are
// the first n / 4, zero behind them, BLINDING_ROWS random rows at the very end
void fill_witness_column(Kind kind, bool blinded, SplitMix& r, int word_bits, size_t n, Limbs* out) {
    const size_t live = n / 4;
    const uint64_t word_mask = word_bits >= 64 ? ~0ull : ((1ull << word_bits) - 1);
    uint64_t even_mask = 0;
    for (int b = 0; b < word_bits; b += 2) even_mask |= 1ull << b;
    for (size_t i = 0; i < n; ++i) out[i] = Limbs{0, 0, 0, 0};
    if (kind == Kind::Full) { for (size_t i = 0; i < n; ++i) out[i] = r.element(); return; }
    for (size_t i = 0; i < live; ++i) {
        const uint64_t v = r.next();
        out[i][0] = kind == Kind::Flag ? (v & 1) : kind == Kind::Word ? (v & word_mask) : kind == Kind::Even ? (v & even_mask) : (v & ((1ull << (word_bits / 2)) - 1));
    }
    if (kind == Kind::Sorted) std::sort(out, out + live, [](const Limbs& a, const Limbs& b) { return a[0] < b[0]; });
    if (blinded) for (size_t i = n - BLINDING_ROWS; i < n; ++i) out[i] = r.element();
}

// synthetic gate set with the shape of the reference's (selector-gated constraints up to degree 6)
std::vector<Expr> synthetic_gates(Field f, uint32_t n_advice, uint32_t n_fixed, int n_gates) {
    SplitMix r{0x6a7e};
    auto adv = [&]() { const uint32_t c = (uint32_t)(r.next() % n_advice); const int rot[5] = {0, 0, 0, 1, -1}; return advice(c, rot[r.next() % 5]); };
    const Expr one = constant(host::one(f)), two = constant(host::from_u64(f, 2));
    std::vector<Expr> gates;
    for (int g = 0; g < n_gates; ++g) {
        const Expr sel = selector((uint32_t)(r.next() % n_fixed));
        Expr body;
        switch (g % 4) {
            case 0: body = adv() + scaled(adv(), host::from_u64(f, 1u << 16)) - adv(); break;
            case 1: body = adv() * adv() - adv(); break;
            case 2: { const Expr v = adv(); body = v * (one - v) * (two - v); break; }
            default: { const Expr a = adv(), b = adv(); body = (a * a - adv()) * (b * b - adv()) * (adv() - one); }
        }
        gates.push_back(sel * body);
    }
    return gates;
}

// host evaluation of one row (the check of the device evaluator)
// n: rows of one cyclic domain (a power of two); rows beyond n belong to further blocks of n rows each (the coset-block layout):
// a rotation stays inside its block
Limbs eval_host(Field f, const Expr& e, const std::vector<std::vector<Limbs>>& cols, const Program& p, size_t row, size_t n, size_t rot_step) {
    switch (e->kind) {
        case Expression::Constant: return e->value;
        case Expression::Negated: return host::neg(f, eval_host(f, e->a, cols, p, row, n, rot_step));
        case Expression::Scaled: return host::mul(f, eval_host(f, e->a, cols, p, row, n, rot_step), e->value);
        case Expression::Sum: return host::add(f, eval_host(f, e->a, cols, p, row, n, rot_step), eval_host(f, e->b, cols, p, row, n, rot_step));
        case Expression::Product: return host::mul(f, eval_host(f, e->a, cols, p, row, n, rot_step), eval_host(f, e->b, cols, p, row, n, rot_step));
        default: {
            size_t slot = 0;
            while (!(p.columns[slot].first == e->kind && p.columns[slot].second == e->column)) ++slot;
            const size_t r = (row & ~(n - 1)) | ((row + (size_t)((long long)e->rotation * (long long)rot_step)) & (n - 1));
            return cols[slot][r];
        }
    }
}

// c[i] = a[i] * f[i % period] over `threads` host threads: the pointwise steps EvaluationDomain does on the Rust host between
// its best_fft calls (x n^-1 after the inverse transform, the zeta shift before the coset transform)
void host_scale(Field f, Limbs* a, size_t n, const Limbs* factors, size_t period) {
    const unsigned T = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([=] { for (size_t i = n * t / T; i < n * (t + 1) / T; ++i) a[i] = host::mul(f, a[i], factors[i % period]); });
    for (auto& t : th) t.join();
}

// --mode dropin / dropin-batched: every polynomial lives in HOST memory (separate allocations, as the Rust prover's Vec<F>s) and
// crosses PCIe inside the libtrh calls; see tiny-ram-halo2_amd/replay.py::run_dropin for the two levels.  Here the compiled host
// also does EvaluationDomain's pointwise steps itself (timed apart as host_pointwise_ms), so the values ARE the prover's and the two
// levels can be checked against each other: same commitments, same coefficient forms, same extended cosets.
int run_dropin(int word_bits, size_t batch, bool witness, bool batched, int max_columns, bool pinned) {
    init(0);
    const uint32_t k = 2 + word_bits / 2;
    const size_t n = (size_t)1 << k;
    const Curve curve = Curve::Vesta;
    const Field field = scalar_field(curve);
    EvaluationDomain dom(field, QUOTIENT_J, k);
    const uint32_t ek = dom.extended_k;
    const size_t N = dom.extended_len();
    Params params(curve, k, 0x1234567, 0x89abcdef, true);
    int lag_total = N_INSTANCE + N_ADVICE + 3 * N_LOOKUPS + N_PERM_PRODUCTS;
    if (max_columns > 0 && max_columns < lag_total) lag_total = max_columns;
    if (batch > (size_t)lag_total) batch = lag_total;
    std::vector<std::pair<Kind, bool>> kinds;
    for (const ColumnClass& c : WITNESS_CLASSES) for (int i = 0; i < c.count; ++i) kinds.push_back({c.kind, c.blinded});
    const Limbs omega_inv = dom.constant(1), ext_omega = dom.constant(2), ext_omega_inv = dom.constant(3), ifft_div = dom.constant(4), ext_div = dom.constant(5);
    const Limbs zeta = dom.constant(6), zeta2 = dom.constant(7);
    const Limbs into_coset[3] = {host::one(field), zeta, zeta2};
    SplitMix rng{0xc01};
    double ms_commit = 0, ms_intt = 0, ms_ext = 0, ms_commit_coeff = 0, ms_ext_inv = 0, ms_ipa = 0, ms_host = 0;
    std::vector<std::vector<Limbs>> cols(batch, std::vector<Limbs>(n)), exts(batched ? batch : 2, std::vector<Limbs>(N));
    std::vector<Limbs> blinds(batch);
    DeviceBuffer scratch(batch * n * 32);
    // --pinned: the host keeps its polynomials in a page-locked arena (here: the reused column / coset buffers are registered once):
    // the DMA engine then reads and writes them directly, no staging ring, no copy threads
    if (pinned) {
        for (auto& c : cols) check(trh_host_register(c.data(), c.size() * 32), "host_register");
        for (auto& e : exts) check(trh_host_register(e.data(), e.size() * 32), "host_register");
    }
    trh_io_stats_t io0;
    check(trh_io_stats(&io0, 1), "io_stats");
    // cross-check state: the first batch is replayed through BOTH levels
    for (int done = 0; done < lag_total; done += (int)batch) {
        const size_t b = std::min(batch, (size_t)(lag_total - done));
        for (size_t c = 0; c < b; ++c) {
            if (!witness) { for (size_t i = 0; i < n; ++i) cols[c][i] = rng.element(); continue; }
            fill_witness_column(kinds[done + c].first, kinds[done + c].second, rng, word_bits, n, cols[c].data());
            scratch.upload(cols[c].data(), n * 32);  // canonical small integers -> field elements (input preparation, not the replayed path)
            check(trh_field_op_dev((int)field, 6, scratch.data(), nullptr, scratch.data(), n, nullptr), "to_mont");
            scratch.download(cols[c].data(), n * 32);
        }
        for (size_t i = 0; i < b; ++i) blinds[i] = rng.element();
        std::vector<Point> pts(b);
        const bool cross = done == 0;
        std::vector<std::vector<Limbs>> keep;  // the first three Lagrange columns, for the cross-check of the other level
        if (cross) for (size_t c = 0; c < std::min<size_t>(3, b); ++c) keep.push_back(cols[c]);
        if (!batched) {
            for (size_t c = 0; c < b; ++c) {
                std::vector<Limbs> sc(cols[c]);  // poly || blind: the Rust side chains the two slices, the copy is not the library's time
                sc.push_back(blinds[c]);
                double t0 = now_ms();
                pts[c] = params.g_lagrange().msm(sc);
                ms_commit += now_ms() - t0;
                t0 = now_ms();
                best_fft(field, cols[c], omega_inv, k);
                ms_intt += now_ms() - t0;
                t0 = now_ms();
                host_scale(field, cols[c].data(), n, &ifft_div, 1);
                std::vector<Limbs>& e = exts[c & 1];
                std::fill(e.begin() + n, e.end(), Limbs{0, 0, 0, 0});
                std::copy(cols[c].begin(), cols[c].end(), e.begin());
                host_scale(field, e.data(), n, into_coset, 3);
                ms_host += now_ms() - t0;
                t0 = now_ms();
                best_fft(field, e, ext_omega, ek);
                ms_ext += now_ms() - t0;
                if (cross && c < keep.size()) {  // the batched level on the same column: identical point, coefficients and coset
                    std::vector<Limbs> a = keep[c], x(N);
                    std::vector<const std::vector<Limbs>*> in1{&a};
                    std::vector<std::vector<Limbs>*> io1{&a}, out1{&x};
                    const Point p2 = params.commit_lagrange_batch_host(in1, {blinds[c]})[0];
                    expect(std::memcmp(&p2, &pts[c], sizeof(Point)) == 0, "literal commit_lagrange == trh_commit_batch_host");
                    dom.lagrange_to_coeff_host(io1);
                    expect(a == cols[c], "best_fft + host x n^-1 == trh_domain_lagrange_to_coeff_host");
                    dom.coeff_to_extended_host(in1, out1);
                    expect(x == e, "host zero-pad + zeta shift + best_fft == trh_domain_coeff_to_extended_host");
                }
            }
        } else {
            std::vector<const std::vector<Limbs>*> in;
            std::vector<std::vector<Limbs>*> io, out;
            for (size_t c = 0; c < b; ++c) { in.push_back(&cols[c]); io.push_back(&cols[c]); out.push_back(&exts[c]); }
            double t0 = now_ms();
            pts = params.commit_lagrange_batch_host(in, std::vector<Limbs>(blinds.begin(), blinds.begin() + b));
            double t1 = now_ms();
            dom.lagrange_to_coeff_host(io);
            double t2 = now_ms();
            dom.coeff_to_extended_host(in, out);
            double t3 = now_ms();
            ms_commit += t1 - t0; ms_intt += t2 - t1; ms_ext += t3 - t2;
            if (cross) {  // the literal level on the first column
                std::vector<Limbs> a = keep[0];
                const Point p1 = params.commit_lagrange(a, blinds[0]);
                expect(std::memcmp(&p1, &pts[0], sizeof(Point)) == 0, "trh_commit_batch_host == literal commit_lagrange");
                best_fft(field, a, omega_inv, k);
                host_scale(field, a.data(), n, &ifft_div, 1);
                expect(a == cols[0], "trh_domain_lagrange_to_coeff_host == best_fft + host x n^-1");
                std::vector<Limbs> e(N, Limbs{0, 0, 0, 0});
                std::copy(a.begin(), a.end(), e.begin());
                host_scale(field, e.data(), n, into_coset, 3);
                best_fft(field, e, ext_omega, ek);
                expect(e == exts[0], "trh_domain_coeff_to_extended_host == host zero-pad + zeta shift + best_fft");
            }
        }
    }
    // coefficient-basis commits (random vanishing polynomial + h pieces)
    const size_t ncoef = 1 + N_H_PIECES;
    std::vector<std::vector<Limbs>> cf(ncoef, std::vector<Limbs>(n));
    for (auto& c : cf) for (auto& v : c) v = rng.element();
    std::vector<Limbs> bl(ncoef);
    for (auto& v : bl) v = rng.element();
    {
        const double t0 = now_ms();
        if (!batched) for (size_t c = 0; c < ncoef; ++c) (void)params.commit(cf[c], bl[c]);
        else { std::vector<const std::vector<Limbs>*> in; for (auto& c : cf) in.push_back(&c); (void)params.commit_batch_host(in, bl); }
        ms_commit_coeff = now_ms() - t0;
    }
    // h(X): 2^extended_k values back from the host's gate evaluation
    {
        std::vector<Limbs> h(N);
        for (auto& v : h) v = rng.element();
        std::vector<Limbs> h2 = h;
        const double t0 = now_ms();
        if (!batched) best_fft(field, h, ext_omega_inv, ek); else dom.extended_to_coeff_host(h, false);
        ms_ext_inv = now_ms() - t0;
        // cross-check: literal + the host's pointwise tail == the domain form
        const Limbs from_div[3] = {ext_div, host::mul(field, ext_div, zeta2), host::mul(field, ext_div, zeta)};
        if (!batched) { host_scale(field, h.data(), N, from_div, 3); dom.extended_to_coeff_host(h2, false); }
        else { best_fft(field, h2, ext_omega_inv, ek); host_scale(field, h2.data(), N, from_div, 3); }
        expect(h == h2, "extended_to_coeff: best_fft + host tail == trh_domain_extended_to_coeff_host");
    }
    // the opening
    {
        std::vector<Limbs> p(n), sp(n);
        for (auto& v : p) v = rng.element();
        for (auto& v : sp) v = rng.element();
        const double t0 = now_ms();
        if (!batched) {  // commitment::create_proof on the host: S, then two best_multiexp per round over the folded generators (host bases)
            (void)params.commit(sp, rng.element());
            const std::vector<Affine> g = params.g().download();
            for (uint32_t j = 0; j < k; ++j) {
                const size_t half = (size_t)1 << (k - j - 1);
                (void)best_multiexp(curve, std::vector<Limbs>(p.begin() + half, p.begin() + 2 * half), std::vector<Affine>(g.begin(), g.begin() + half));
                (void)best_multiexp(curve, std::vector<Limbs>(p.begin(), p.begin() + half), std::vector<Affine>(g.begin() + half, g.begin() + 2 * half));
            }
        } else {
            DeviceBuffer pd(n * 32), sd(n * 32);
            pd.upload(p.data(), n * 32); sd.upload(sp.data(), n * 32);
            Transcript tr;
            trh_transcript_t tcb{&tr, tr_write_point, tr_write_scalar, tr_squeeze};
            SplitMix prng{0x99};
            (void)ipa_create_proof(params, pd, rng.element(), rng.element(), sd, rng.element(), tcb, rng_scalar, &prng);
        }
        ms_ipa = now_ms() - t0;
    }
    trh_io_stats_t io;
    check(trh_io_stats(&io, 0), "io_stats");
    if (pinned) {
        for (auto& c : cols) (void)trh_host_unregister(c.data());
        for (auto& e : exts) (void)trh_host_unregister(e.data());
    }
    const double total = ms_commit + ms_intt + ms_ext + ms_commit_coeff + ms_ext_inv + ms_ipa;
    std::printf("{\"driver\": \"examples/replay.cpp\", \"mode\": \"%s\", \"word_bits\": %d, \"k\": %u, \"batch\": %zu, \"columns\": \"%s\", \"columns_replayed\": %d, \"checks_failed\": %d, "
                "\"wall_ms_incl_pcie\": {\"commit_lagrange\": %.3f, \"lagrange_to_coeff\": %.3f, \"coeff_to_extended\": %.3f, \"commit\": %.3f, \"extended_to_coeff\": %.3f, \"ipa\": %.3f}, "
                "\"wall_ms_incl_pcie_total\": %.3f, \"host_pointwise_ms\": %.3f, \"pcie\": {\"h2d_GB\": %.3f, \"d2h_GB\": %.3f, \"h2d_GBps_in_copies\": %.2f, \"d2h_GBps_in_copies\": %.2f, "
                "\"GBps_over_call_time\": %.2f, \"link_peak_GBps_per_direction\": 57.0}}\n",
                pinned ? (batched ? "dropin-batched-pinned" : "dropin-literal-pinned") : (batched ? "dropin-batched" : "dropin-literal"), word_bits, k, batch, witness ? "witness" : "random", lag_total, failures, ms_commit, ms_intt, ms_ext, ms_commit_coeff, ms_ext_inv, ms_ipa, total,
                ms_host, io.h2d_bytes / 1e9, io.d2h_bytes / 1e9, io.h2d_bytes / std::max(io.h2d_seconds, 1e-9) / 1e9, io.d2h_bytes / std::max(io.d2h_seconds, 1e-9) / 1e9,
                (io.h2d_bytes + io.d2h_bytes) / std::max(total * 1e-3, 1e-9) / 1e9);
    trh_shutdown();
    return failures ? 1 : 0;
}

}  // namespace

// --devices d0,d1,...: the per-column phase of create_proof (commit_lagrange, lagrange_to_coeff, coeff_to_extended as coset blocks, the
// evaluations at x) COLUMN-SHARDED over the listed devices from this one process -- per entry one host thread, one trh::Context, one Params
// copy and one EvaluationDomain on that device; thread g takes the contiguous columns [lo_g, hi_g) (the split of sharded.shard_range), no
// data-path collective.  A device may be listed twice (two contexts on one chip).  The commitments gathered in column order must equal the
// single-context run's bit for bit (exit code 1 otherwise); per-device times and the single-context time of the same phase are printed.
// The reference proves in one process (/root/reference/src/test_utils.rs:37-54): this is what that process does with a node's GPUs.
int run_sharded(int word_bits, size_t batch, bool witness, const std::vector<int>& devices, int max_columns) {
    init(devices[0]);
    const uint32_t k = 2 + word_bits / 2;
    const size_t n = (size_t)1 << k;
    const Curve curve = Curve::Vesta;
    const Field field = scalar_field(curve);
    int lag_total = N_INSTANCE + N_ADVICE + 3 * N_LOOKUPS + N_PERM_PRODUCTS;
    if (max_columns > 0 && max_columns < lag_total) lag_total = max_columns;
    std::vector<std::pair<Kind, bool>> kinds;
    for (const ColumnClass& c : WITNESS_CLASSES) for (int i = 0; i < c.count; ++i) kinds.push_back({c.kind, c.blinded});
    const size_t G = devices.size();
    const Limbs x_eval = SplitMix{0xe7a}.element();
    struct Shard {
        int device = 0, lo = 0, hi = 0;
        std::unique_ptr<Context> ctx;
        std::unique_ptr<Params> params;
        std::unique_ptr<EvaluationDomain> dom;
        std::unique_ptr<DeviceBuffer> cols, work, ext;
        std::vector<Limbs> blinds;
        std::vector<Point> points;
        std::vector<Limbs> evals;
        double ms[4] = {0, 0, 0, 0}, wall_ms = 0;
        std::string err;
    };
    std::vector<Shard> sh(G);
    const int base = lag_total / (int)G, extra = lag_total % (int)G;
    for (size_t g = 0; g < G; ++g) {
        sh[g].device = devices[g];
        sh[g].lo = (int)g * base + std::min((int)g, extra);
        sh[g].hi = sh[g].lo + base + ((int)g < extra ? 1 : 0);
    }
    // the per-column steps over `count` columns of `buf` (in place), batch by batch, on the bound context's stream
    auto phase = [&](Shard& s, DeviceBuffer& buf, size_t count, const Limbs* blinds, std::vector<Point>& pts, std::vector<Limbs>& evals, double* ms) {
        void* st = s.ctx->stream();
        const uint32_t D = s.dom->quotient_blocks();
        pts.clear(); evals.clear();
        for (size_t done = 0; done < count; done += batch) {
            const size_t b = std::min(batch, count - done);
            void* c0 = buf.at(done * n * 32);
            double t0 = now_ms();
            std::vector<Point> p(b);
            check(trh_commit_batch_dev(s.params->g_lagrange().handle(), c0, n, b, (const uint64_t*)(blinds + done), st, (uint64_t*)p.data()), "commit_batch");
            double t1 = now_ms();
            s.dom->lagrange_to_coeff(c0, b, st);
            check(trh_stream_synchronize(st), "sync");
            double t2 = now_ms();
            s.dom->coeff_to_extended_blocks(c0, s.ext->data(), b, D, st);
            check(trh_stream_synchronize(st), "sync");
            double t3 = now_ms();
            const std::vector<Limbs> e = eval_polynomials(field, c0, n, b, x_eval, st);
            double t4 = now_ms();
            if (ms) { ms[0] += t1 - t0; ms[1] += t2 - t1; ms[2] += t3 - t2; ms[3] += t4 - t3; }
            pts.insert(pts.end(), p.begin(), p.end());
            evals.insert(evals.end(), e.begin(), e.end());
        }
    };
    auto setup = [&](Shard& s) {
        s.ctx.reset(new Context(s.device));
        s.ctx->bind();
        s.params.reset(new Params(curve, k, 0x1234567, 0x89abcdef, true));
        s.dom.reset(new EvaluationDomain(field, QUOTIENT_J, k));
        s.params->reserve(batch);
        s.dom->reserve(batch);
        const size_t cnt = (size_t)(s.hi - s.lo);
        s.cols.reset(new DeviceBuffer(std::max<size_t>(cnt, 1) * n * 32));
        s.work.reset(new DeviceBuffer(std::max<size_t>(cnt, 1) * n * 32));
        s.ext.reset(new DeviceBuffer(std::min(batch, std::max<size_t>(cnt, 1)) * s.dom->quotient_blocks() * n * 32));
        std::vector<Limbs> host(n);
        for (size_t c = 0; c < cnt; ++c) {  // every column has its own generator state: the single-context run rebuilds exactly these
            SplitMix rng{0xc01 + (uint64_t)(s.lo + (int)c) * 7919};
            if (witness) fill_witness_column(kinds[s.lo + c].first, kinds[s.lo + c].second, rng, word_bits, n, host.data());
            else for (size_t i = 0; i < n; ++i) host[i] = rng.element();
            s.cols->upload(host.data(), n * 32, c * n * 32);
        }
        if (witness && cnt) check(trh_field_op_dev((int)field, 6 /* to_montgomery */, s.cols->data(), nullptr, s.cols->data(), cnt * n, nullptr), "to_mont");
        check(trh_stream_synchronize(nullptr), "sync");
        s.blinds.resize(cnt);
        for (size_t c = 0; c < cnt; ++c) s.blinds[c] = SplitMix{0xb11d + (uint64_t)(s.lo + (int)c)}.element();
        Context::unbind();
    };
    auto copy_cols = [&](Shard& s) {  // work <- cols (one kernel per column: the ABI's copy helpers would bounce over the host)
        const std::vector<Limbs> unit(1, host::one(field));
        for (int c = 0; c < s.hi - s.lo; ++c) lincomb(field, s.cols->at((size_t)c * n * 32), n, unit, s.work->at((size_t)c * n * 32), s.ctx->stream());
        check(trh_stream_synchronize(s.ctx->stream()), "sync");
    };
    for (Shard& s : sh) setup(s);
    // single-context reference: shard 0's context takes every column, step by step (the other shards' columns are rebuilt on its device)
    std::vector<Point> ref_pts;
    std::vector<Limbs> ref_evals;
    double single_ms[4] = {0, 0, 0, 0}, single_wall = 0;
    {
        Shard& r = sh[0];
        r.ctx->bind();
        DeviceBuffer all((size_t)lag_total * n * 32);
        std::vector<Limbs> host(n), blinds(lag_total);
        for (int c = 0; c < lag_total; ++c) {
            SplitMix rng{0xc01 + (uint64_t)c * 7919};
            if (witness) fill_witness_column(kinds[c].first, kinds[c].second, rng, word_bits, n, host.data());
            else for (size_t i = 0; i < n; ++i) host[i] = rng.element();
            all.upload(host.data(), n * 32, (size_t)c * n * 32);
            blinds[c] = SplitMix{0xb11d + (uint64_t)c}.element();
        }
        if (witness) check(trh_field_op_dev((int)field, 6, all.data(), nullptr, all.data(), (size_t)lag_total * n, nullptr), "to_mont");
        check(trh_stream_synchronize(nullptr), "sync");
        {   // warm-up at the real batch size (scratch, tables)
            DeviceBuffer warm(std::min(batch, (size_t)lag_total) * n * 32);
            const std::vector<Limbs> unit(1, host::one(field));
            for (size_t c = 0; c < std::min(batch, (size_t)lag_total); ++c) lincomb(field, all.at(c * n * 32), n, unit, warm.at(c * n * 32), r.ctx->stream());
            std::vector<Point> p; std::vector<Limbs> e;
            phase(r, warm, std::min(batch, (size_t)lag_total), blinds.data(), p, e, nullptr);
        }
        const double t0 = now_ms();
        phase(r, all, (size_t)lag_total, blinds.data(), ref_pts, ref_evals, single_ms);
        single_wall = now_ms() - t0;
        Context::unbind();
    }
    // the sharded run: one thread per device, started together
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> ths;
    for (size_t g = 0; g < G; ++g) ths.emplace_back([&, g] {
        Shard& s = sh[g];
        try {
            s.ctx->bind();
            const size_t cnt = (size_t)(s.hi - s.lo);
            copy_cols(s);
            { std::vector<Point> p; std::vector<Limbs> e; phase(s, *s.work, std::min(batch, cnt), s.blinds.data(), p, e, nullptr); }  // warm-up
            copy_cols(s);
            ready.fetch_add(1);
            while (!go.load()) std::this_thread::yield();
            const double t0 = now_ms();
            phase(s, *s.work, cnt, s.blinds.data(), s.points, s.evals, s.ms);
            s.wall_ms = now_ms() - t0;
            Context::unbind();
        } catch (const std::exception& e) { s.err = e.what(); ready.fetch_add(1); }
    });
    while (ready.load() < (int)G) std::this_thread::yield();
    const double t0 = now_ms();
    go.store(true);
    for (auto& t : ths) t.join();
    const double wall = now_ms() - t0;
    bool identical = true;
    for (Shard& s : sh) {
        if (!s.err.empty()) { std::fprintf(stderr, "shard on device %d failed: %s\n", s.device, s.err.c_str()); return 2; }
        for (int c = s.lo; c < s.hi; ++c) {
            identical = identical && std::memcmp(&s.points[c - s.lo], &ref_pts[c], sizeof(Point)) == 0 && s.evals[c - s.lo] == ref_evals[c];
        }
    }
    std::printf("{\"mode\": \"column-sharded\", \"k\": %u, \"columns\": \"%s\", \"devices\": [", k, witness ? "witness" : "random");
    for (size_t g = 0; g < G; ++g) std::printf("%s%d", g ? ", " : "", devices[g]);
    std::printf("], \"columns_replayed\": %d, \"per_device\": [", lag_total);
    for (size_t g = 0; g < G; ++g)
        std::printf("%s{\"device\": %d, \"columns\": [%d, %d], \"wall_ms\": %.3f, \"ms\": {\"commit_lagrange\": %.3f, \"lagrange_to_coeff\": %.3f, \"coeff_to_extended\": %.3f, \"evals\": %.3f}}",
                    g ? ", " : "", sh[g].device, sh[g].lo, sh[g].hi, sh[g].wall_ms, sh[g].ms[0], sh[g].ms[1], sh[g].ms[2], sh[g].ms[3]);
    std::printf("], \"wall_ms\": %.3f, \"single_context\": {\"wall_ms\": %.3f, \"ms\": {\"commit_lagrange\": %.3f, \"lagrange_to_coeff\": %.3f, \"coeff_to_extended\": %.3f, \"evals\": %.3f}}, "
                "\"commitments_identical_to_single_context\": %s}\n", wall, single_wall, single_ms[0], single_ms[1], single_ms[2], single_ms[3], identical ? "true" : "false");
    for (Shard& s : sh) {  // everything a shard owns lives on its context's device: released while it is bound
        s.ctx->bind();
        s.ext.reset(); s.work.reset(); s.cols.reset(); s.dom.reset(); s.params.reset();
        Context::unbind();
    }
    return identical ? 0 : 1;
}

int main(int argc, char** argv) {
    int word_bits = 32;
    size_t batch = 64;
    bool witness = false;
    std::string mode = "resident";
    int max_columns = 0;
    bool overlap = false, pinned = false;
    std::vector<int> devices;
    for (int i = 1; i < argc; ++i) if (std::string(argv[i]) == "--overlap") { overlap = true; for (int q = i; q + 1 < argc; ++q) argv[q] = argv[q + 1]; --argc; break; }
    for (int i = 1; i < argc; ++i) if (std::string(argv[i]) == "--host-clock") { g_host_clock = true; for (int q = i; q + 1 < argc; ++q) argv[q] = argv[q + 1]; --argc; break; }
    for (int i = 1; i < argc; ++i) if (std::string(argv[i]) == "--pinned") { pinned = true; for (int q = i; q + 1 < argc; ++q) argv[q] = argv[q + 1]; --argc; break; }
    for (int i = 1; i + 1 < argc; i += 2) {
        if (std::string(argv[i]) == "--word-bits") word_bits = std::atoi(argv[i + 1]);
        else if (std::string(argv[i]) == "--batch") batch = (size_t)std::atol(argv[i + 1]);
        else if (std::string(argv[i]) == "--columns") witness = std::string(argv[i + 1]) == "witness";
        else if (std::string(argv[i]) == "--mode") mode = argv[i + 1];
        else if (std::string(argv[i]) == "--max-columns") max_columns = std::atoi(argv[i + 1]);
        else if (std::string(argv[i]) == "--devices") { for (const char* q = argv[i + 1]; *q;) { devices.push_back(std::atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; } }
    }
    try {
        if (!devices.empty()) return run_sharded(word_bits, batch, witness, devices, max_columns);
        if (mode == "dropin" || mode == "dropin-batched") return run_dropin(word_bits, batch, witness, mode == "dropin-batched", max_columns, pinned);
        init(0);
        const uint32_t k = 2 + word_bits / 2;
        const size_t n = (size_t)1 << k;
        const Curve curve = Curve::Vesta;  // the reference proves over Fp with Params<EqAffine> (test_utils.rs:8, 21)
        const Field field = scalar_field(curve);
        EvaluationDomain dom(field, QUOTIENT_J, k);
        const uint32_t ek = dom.extended_k;
        const size_t N = dom.extended_len();
        require(ek == k + 3, "extended_k == k + 3");

        Timer t_setup;
        Params params(curve, k, 0x1234567, 0x89abcdef, true);
        double setup_ms = t_setup.stop();
        // HBM of the fixed-base tables the three resident sets carry (W windows x points x 128-byte records each)
        double tables_gb = 0;
        for (const Bases* b : {&params.g(), &params.g_lagrange(), &params.ipa_bases()}) {
            const int c = trh_bases_precomputed_window_bits(b->handle());
            if (c > 0 && (b != &params.ipa_bases() || b->handle() != params.g().handle())) tables_gb += (double)(255 / c + 1) * (double)trh_bases_len(b->handle()) * 128.0 / 1e9;
        }

        const int lag_total = N_INSTANCE + N_ADVICE + 3 * N_LOOKUPS + N_PERM_PRODUCTS;
        if (batch > (size_t)lag_total) batch = lag_total;
        // the extended domain as the QUOTIENT_J - 1 = 5 coset blocks (of 8) the quotient needs: D * n rows per column (trh.h)
        const uint32_t D = dom.quotient_blocks();
        const size_t EN = (size_t)D * n;
        {   // keygen-time sizing: the first proof of the process then allocates and builds nothing inside its steps
            Timer t_reserve;
            params.reserve(batch);
            dom.reserve(batch);
            setup_ms += t_reserve.stop();
        }
        DeviceBuffer cols(batch * n * 32), ext(batch * EN * 32), h_num(EN * 32);
        std::vector<Limbs> host_cols(batch * n), blinds(batch);
        SplitMix rng{0xc01};
        // the coefficient forms stay resident for the multiopen argument (497 + 6 polynomials)
        DeviceBuffer coeff_all((size_t)(lag_total + 1 + N_H_PIECES) * n * 32);
        std::vector<std::pair<Kind, bool>> kinds;
        for (const ColumnClass& c : WITNESS_CLASSES) for (int i = 0; i < c.count; ++i) kinds.push_back({c.kind, c.blinded});
        require((int)kinds.size() == lag_total, "column classes cover the 497 columns");
        auto upload_columns = [&](int first, size_t b) {  // this batch's columns -> `cols` (Montgomery form), host copy in host_cols
            if (!witness) {
                for (size_t i = 0; i < b * n; ++i) host_cols[i] = rng.element();
                cols.upload(host_cols.data(), b * n * 32);
                return;
            }
            for (size_t c = 0; c < b; ++c) fill_witness_column(kinds[first + c].first, kinds[first + c].second, rng, word_bits, n, host_cols.data() + c * n);
            cols.upload(host_cols.data(), b * n * 32);
            check(trh_field_op_dev((int)field, 6 /* to_montgomery */, cols.data(), nullptr, cols.data(), b * n, nullptr), "to_mont");
            cols.download(host_cols.data(), b * n * 32);  // the checks below compare against what the device was given
        };
        const Limbs x_eval = rng.element();
        double ms_commit = 0, ms_intt = 0, ms_ext = 0, ms_evals = 0, ms_h = 0, ms_commit_coeff = 0, ms_ext_inv = 0, ms_ipa = 0, ms_lookup = 0;
        {   // lookup argument: permuted columns of the 31 lookups; inputs drawn from the table's values
            const size_t distinct = std::min(n, (size_t)1 << 16);
            std::vector<Limbs> vals(distinct), table(n), input(n);
            SplitMix lr{0x7ab1e};
            for (auto& v : vals) v = lr.element();
            DeviceBuffer d_in((size_t)N_LOOKUPS * n * 32), d_tab((size_t)N_LOOKUPS * n * 32), d_pa((size_t)N_LOOKUPS * n * 32), d_ps((size_t)N_LOOKUPS * n * 32);
            std::vector<Limbs> in0, tab0;
            for (int li = 0; li < N_LOOKUPS; ++li) {
                for (size_t i = 0; i < n; ++i) { table[i] = vals[i % distinct]; input[i] = vals[lr.next() % distinct]; }
                d_in.upload(input.data(), n * 32, (size_t)li * n * 32); d_tab.upload(table.data(), n * 32, (size_t)li * n * 32);
                if (li == 0) { in0 = input; tab0 = table; }
            }
            lookup_permute_batch(field, d_in.data(), d_tab.data(), n, n, N_LOOKUPS, d_pa.data(), d_ps.data());  // untimed first call: the scratch of the sort is allocated once per process
            Timer tl;
            lookup_permute_batch(field, d_in.data(), d_tab.data(), n, n, N_LOOKUPS, d_pa.data(), d_ps.data());  // the 31 lookups of the proof in one call
            ms_lookup += tl.stop();
            {   // the argument's defining constraints on lookup 0: A'[i] == S'[i] or A'[i] == A'[i-1]; A' is a permutation of A (sums agree)
                std::vector<Limbs> pa(n), ps(n);
                d_pa.download(pa.data(), n * 32); d_ps.download(ps.data(), n * 32);
                bool ok = true;
                Limbs sa{0, 0, 0, 0}, sb{0, 0, 0, 0}, st{0, 0, 0, 0}, su{0, 0, 0, 0};
                for (size_t i = 0; i < n; ++i) {
                    ok = ok && (pa[i] == ps[i] || (i > 0 && pa[i] == pa[i - 1]));
                    sa = host::add(field, sa, in0[i]); sb = host::add(field, sb, pa[i]);
                    st = host::add(field, st, tab0[i]); su = host::add(field, su, ps[i]);
                }
                expect(ok && sa == sb && st == su, "lookup permuted columns");
            }
        }
        size_t last_b = 0;
        std::vector<Limbs> first_col, first_coeff;

        // the columns are built first (untimed: witness generation is the host's; with it between the batches the GPU idles and the
        // timed bursts run on ramping clocks) and wait as Lagrange forms in the rows of coeff_all
        const std::vector<Limbs> unit1(1, host::one(field));
        for (int done = 0; done < lag_total; done += (int)batch) {
            const size_t b = std::min(batch, (size_t)(lag_total - done));
            upload_columns(done, b);
            if (done == 0) first_col.assign(host_cols.begin(), host_cols.begin() + n);
            for (size_t c = 0; c < b; ++c) lincomb(field, cols.at(c * n * 32), n, unit1, coeff_all.at(((size_t)done + c) * n * 32));
        }
        check(trh_stream_synchronize(nullptr), "sync");
        for (int done = 0; done < lag_total; done += (int)batch) {
            const size_t b = std::min(batch, (size_t)(lag_total - done));
            for (size_t c = 0; c < b; ++c) lincomb(field, coeff_all.at(((size_t)done + c) * n * 32), n, unit1, cols.at(c * n * 32));
            check(trh_stream_synchronize(nullptr), "sync");
            for (size_t i = 0; i < b; ++i) blinds[i] = rng.element();
            Timer t1;
            const std::vector<Point> pts = params.commit_lagrange_batch(cols, b, std::vector<Limbs>(blinds.begin(), blinds.begin() + b));
            { const double dt = t1.stop(); ms_commit += dt; if (std::getenv("TRH_REPLAY_VERBOSE")) std::fprintf(stderr, "batch at %d: commit %.2f ms\n", done, dt); }
            if (done == 0) {  // Params::commit_lagrange of column 0 through the host-scalar path must give the same point
                const Point single = params.commit_lagrange(first_col, blinds[0]);
                expect(std::memcmp(&single, &pts[0], sizeof(Point)) == 0, "commit_lagrange_batch[0] == commit_lagrange");
            }
            Timer t2;
            dom.lagrange_to_coeff(cols.data(), b);
            ms_intt += t2.stop();
            if (done == 0) { first_coeff.resize(n); cols.download(first_coeff.data(), n * 32); }
            Timer t3;
            dom.coeff_to_extended_blocks(cols.data(), ext.data(), b, D);
            ms_ext += t3.stop();
            Timer t4;
            std::vector<Limbs> evals;
            evals = eval_polynomials(field, cols.data(), n, b, x_eval);
            ms_evals += t4.stop();
            if (done == 0) {  // eval_polynomial: Horner on the host
                Limbs acc{0, 0, 0, 0};
                for (size_t i = n; i-- > 0;) acc = host::add(field, host::mul(field, acc, x_eval), first_coeff[i]);
                expect(acc == evals[0], "eval_polynomial(column 0, x)");
            }
            {   // keep the coefficient forms for the multiopen argument (device-to-device through the ABI's copy helpers would bounce over the
                // host: one kernel per column instead); behind the batch's timed steps
                std::vector<Limbs> unit(1, host::one(field));
                for (size_t c = 0; c < b; ++c) lincomb(field, cols.at(c * n * 32), n, unit, coeff_all.at(((size_t)done + c) * n * 32));
                check(trh_stream_synchronize(nullptr), "sync");
            }
            last_b = b;
        }

        // h(X) numerator over the last batch of extended cosets (synthetic gate set, see replay.py)
        const uint32_t nres = (uint32_t)last_b;
        const std::vector<Expr> gates = synthetic_gates(field, nres > 4 ? nres - 4 : 1, nres < 4 ? nres : 4, N_SYNTH_GATES);
        const Limbs y = host::from_u64(field, 0x5eed);
        GateEvaluator gev(compile_gates(field, gates, y));
        std::vector<const void*> res(gev.program.columns.size());
        for (size_t i = 0; i < res.size(); ++i) res[i] = ext.at((i % nres) * EN * 32);
        Timer t5;
        gev.eval_blocks(res, h_num.data(), k, D);
        ms_h = t5.stop();
        {   // three rows against the host evaluation (rotations stay inside their block)
            std::vector<std::vector<Limbs>> hc(res.size(), std::vector<Limbs>(EN));
            for (size_t i = 0; i < res.size(); ++i) check(trh_memcpy_d2h(hc[i].data(), res[i], EN * 32), "d2h");
            std::vector<Limbs> got(EN);
            h_num.download(got.data(), EN * 32);
            for (size_t row : {(size_t)0, EN - 1, (size_t)2 * n + 4097 % n}) {
                Limbs acc{0, 0, 0, 0};
                for (const Expr& g : gates) acc = host::add(field, host::mul(field, acc, y), eval_host(field, g, hc, gev.program, row, n, 1));
                expect(acc == got[row], "h(X) numerator row");
            }
        }

        {   // permutation argument: the 47 product columns of 4 columns each; here one, over the identity permutation
            // (sigma_j = delta^j omega^i), for which every factor is 1 and z must stay at z[0] = 1 on every row
            const Limbs delta = [&] { Limbs d = host::from_u64(field, 5); for (int i = 0; i < 32; ++i) d = host::mul(field, d, d); return d; }();  // 5^(2^32)
            const Limbs beta = rng.element(), gamma = rng.element();
            const Expr x = fixed(0);
            Expr num, den;
            Limbs dj = host::one(field);
            for (uint32_t j = 0; j < 4; ++j) {
                const Expr tn = advice(j) + scaled(x, host::mul(field, beta, dj)) + constant(gamma);
                const Expr td = advice(j) + scaled(advice(4 + j), beta) + constant(gamma);
                num = num ? num * tn : tn;
                den = den ? den * td : td;
                dj = host::mul(field, dj, delta);
            }
            GrandProduct gp(field, k, num, den);
            DeviceBuffer omegas(n * 32), sig(4 * n * 32), zcol(n * 32);
            check(trh_field_powers_dev((int)field, omegas.data(), n, dom.get_omega().data(), nullptr), "powers");
            Limbs dpow = host::one(field);
            for (uint32_t j = 0; j < 4; ++j) {  // sigma_j = delta^j * omega^i
                check(trh_memcpy_d2h(host_cols.data(), omegas.data(), n * 32), "d2h");
                for (size_t i = 0; i < n; ++i) host_cols[i] = host::mul(field, host_cols[i], dpow);
                sig.upload(host_cols.data(), n * 32, j * n * 32);
                dpow = host::mul(field, dpow, delta);
            }
            std::vector<const void*> pc;
            for (const auto& c : gp.columns()) {
                if (c.first == Expression::Fixed) pc.push_back(omegas.data());
                else pc.push_back(c.second < 4 ? ext.at(c.second * EN * 32) /* any n values serve as the witness */ : sig.at((c.second - 4) * n * 32));
            }
            Timer tp;
            gp.compute(pc, zcol.data());
            const double ms_perm = tp.stop();
            std::vector<Limbs> zh(n);
            zcol.download(zh.data(), n * 32);
            bool ok = true;
            for (size_t i = 0; i < n; ++i) ok = ok && zh[i] == host::one(field);
            expect(ok, "permutation product over the identity permutation stays at 1");
            if (std::getenv("TRH_REPLAY_VERBOSE")) std::fprintf(stderr, "permutation product column: %.3f ms\n", ms_perm);
            // the same chunk three times plus a lookup-shaped row through the fixed-function form (every product column in one launch)
            std::vector<std::vector<trh_product_term_t>> nrows, drows;
            for (int rep = 0; rep < 3; ++rep) {
                std::vector<trh_product_term_t> nr, dr;
                Limbs dj2 = host::one(field);
                for (uint32_t j = 0; j < 4; ++j) {
                    nr.push_back(product_term(ext.at(j * EN * 32), omegas.data(), host::mul(field, beta, dj2), gamma));
                    dr.push_back(product_term(ext.at(j * EN * 32), sig.at(j * n * 32), beta, gamma));
                    dj2 = host::mul(field, dj2, delta);
                }
                nrows.push_back(nr); drows.push_back(dr);
            }
            // lookup-shaped: (A + beta)(S + gamma) / ((A' + beta)(S' + gamma)) with A' = A, S' = S: stays at 1 as well
            nrows.push_back({product_term(ext.at(0), nullptr, Limbs{0, 0, 0, 0}, beta), product_term(sig.data(), nullptr, Limbs{0, 0, 0, 0}, gamma)});
            drows.push_back({product_term(ext.at(0), nullptr, Limbs{0, 0, 0, 0}, beta), product_term(sig.data(), nullptr, Limbs{0, 0, 0, 0}, gamma)});
            DeviceBuffer zall(4 * n * 32);
            Timer tg;
            grand_products(field, k, nrows, drows, zall.data());
            const double ms_gp = tg.stop();
            bool ok2 = true;
            for (int r = 0; r < 4; ++r) {
                zall.download(zh.data(), n * 32, (size_t)r * n * 32);
                for (size_t i = 0; i < n; ++i) ok2 = ok2 && zh[i] == host::one(field);
            }
            expect(ok2, "grand_products (fixed-function rows) over identity permutations / equal lookups stays at 1");
            if (std::getenv("TRH_REPLAY_VERBOSE")) std::fprintf(stderr, "grand_products, 4 rows: %.3f ms\n", ms_gp);
        }

        // coefficient-basis commits: the random vanishing polynomial and the h pieces
        const size_t ncoef = 1 + N_H_PIECES;
        DeviceBuffer cf(ncoef * n * 32);
        for (size_t i = 0; i < ncoef * n; ++i) host_cols[i % host_cols.size()] = rng.element();
        cf.upload(host_cols.data(), ncoef * n * 32);
        {
            std::vector<Limbs> unit(1, host::one(field));
            for (size_t c = 0; c < ncoef; ++c) lincomb(field, cf.at(c * n * 32), n, unit, coeff_all.at(((size_t)lag_total + c) * n * 32));
        }
        std::vector<Limbs> bl(ncoef);
        for (auto& v : bl) v = rng.element();
        Timer t6;
        params.commit_batch(cf, ncoef, bl);
        ms_commit_coeff = t6.stop();

        // extended iNTT of h(X): divide_by_vanishing_poly + extended_to_coeff; round trip check on one column
        DeviceBuffer h_coeff(EN * 32);
        Timer t7;
        dom.blocks_to_quotient(h_num.data(), h_coeff.data(), true);
        ms_ext_inv = t7.stop();
        {   // both layouts of the extended domain return a polynomial they were given: all 2^extended_k points ...
            DeviceBuffer one_col(n * 32), one_ext(N * 32);
            one_col.upload(first_coeff.data(), n * 32);
            dom.coeff_to_extended(one_col.data(), one_ext.data(), 1);
            std::vector<Limbs> full(N);
            one_ext.download(full.data(), N * 32);
            dom.extended_to_coeff(one_ext.data(), 1);
            std::vector<Limbs> back(N);
            one_ext.download(back.data(), N * 32);
            bool ok = true;
            for (size_t i = 0; i < N; ++i) ok = ok && (i < n ? back[i] == first_coeff[i] : back[i] == Limbs{0, 0, 0, 0});
            expect(ok, "extended_to_coeff(coeff_to_extended(a)) == a || 0");
            // ... and the coset blocks: block r entry q is entry q * 2^(extended_k - k) + r of the full form, and the D blocks give a back
            DeviceBuffer blk(EN * 32), hq(EN * 32);
            dom.coeff_to_extended_blocks(one_col.data(), blk.data(), 1, D);
            std::vector<Limbs> bh(EN);
            blk.download(bh.data(), EN * 32);
            const size_t step = N / n;
            for (size_t r = 0; r < D; ++r) for (size_t q = 0; q < n; q += 997) ok = ok && bh[r * n + q] == full[q * step + r];
            dom.blocks_to_quotient(blk.data(), hq.data(), false);
            hq.download(bh.data(), EN * 32);
            for (size_t i = 0; i < EN; ++i) ok = ok && (i < n ? bh[i] == first_coeff[i] : bh[i] == Limbs{0, 0, 0, 0});
            expect(ok, "coset blocks: entries match the full extended domain, blocks_to_quotient(coeff_to_extended_blocks(a)) == a || 0");
        }

        // poly::multiopen::create_proof, polynomial side: four point sets ({x}, {x, wx}, {x, w^-1 x}, {x, wx, w^last x}) over the resident
        // coefficient forms stored set by set; x1 fold, divisions, x2 fold, commitment of q', evaluations at x3, x4 fold
        double ms_multiopen = 0;
        DeviceBuffer p_poly(n * 32), s_poly(n * 32);
        {
            const size_t total = (size_t)lag_total + ncoef;
            // set sizes as replay.py builds them: 40 rotated advice + 31 lookup products + 1 permutation product at {x, wx}; 31 permuted inputs
            // at {x, w^-1 x}; 46 permutation products at {x, wx, w^last x}; everything else at {x}
            const size_t set_sizes[4] = {total - 72 - 31 - 46, 72, 31, 46};
            const Limbs omega = dom.get_omega();
            const Limbs x = x_eval, xw = host::mul(field, x, omega), xwi = host::mul(field, x, host::inv(field, omega));
            Limbs xlast = x;
            for (size_t i = 0; i + BLINDING_ROWS + 1 < n; ++i) xlast = host::mul(field, xlast, omega);
            const std::vector<std::vector<Limbs>> set_points = {{x}, {x, xw}, {x, xwi}, {x, xw, xlast}};
            SplitMix cr{0x3a1};
            const Limbs x1 = cr.element(), x2 = cr.element(), x3 = cr.element(), x4 = cr.element();
            DeviceBuffer q(4 * n * 32), divided(4 * n * 32), tmp(n * 32), qprime(n * 32), fold_in(5 * n * 32);
            Timer tm;
            size_t row = 0;
            for (int sidx = 0; sidx < 4; ++sidx) {
                std::vector<Limbs> pw(set_sizes[sidx]);  // q = ((p_0 x1 + p_1) x1 + ...): coefficient of p_j is x1^(len - 1 - j)
                Limbs acc = host::one(field);
                for (size_t j = set_sizes[sidx]; j-- > 0;) { pw[j] = acc; acc = host::mul(field, acc, x1); }
                lincomb(field, coeff_all.at(row * n * 32), n, pw, q.at((size_t)sidx * n * 32));
                row += set_sizes[sidx];
                // divide by (X - z) for every point of the set; pad back to n coefficients
                check(trh_memcpy_d2h(host_cols.data(), q.at((size_t)sidx * n * 32), 32), "d2h");  // (orders the default stream)
                const void* cur = q.at((size_t)sidx * n * 32);
                size_t len = n;
                for (const Limbs& z : set_points[sidx]) {
                    KateDivider kd(field, len, z, host::inv(field, z));
                    kd.divide(cur, tmp.data());
                    --len;
                    std::vector<Limbs> unit(1, host::one(field));
                    lincomb(field, tmp.data(), len, unit, divided.at((size_t)sidx * n * 32));
                    cur = divided.at((size_t)sidx * n * 32);
                }
                const std::vector<Limbs> zeros(n - len, Limbs{0, 0, 0, 0});
                check(trh_stream_synchronize(nullptr), "sync");
                check(trh_memcpy_h2d((char*)divided.at((size_t)sidx * n * 32) + len * 32, zeros.data(), (n - len) * 32), "pad");
            }
            std::vector<Limbs> p2(4);
            { Limbs acc = host::one(field); for (int j = 3; j >= 0; --j) { p2[j] = acc; acc = host::mul(field, acc, x2); } }
            lincomb(field, divided.data(), n, p2, qprime.data());
            const Point qc = params.commit_batch(qprime, 1, {cr.element()})[0];
            (void)qc;
            const std::vector<Limbs> ev3 = eval_polynomials(field, q.data(), n, 4, x3);
            (void)ev3;
            // p = ((q' x4 + q_0) x4 + q_1) ...
            std::vector<Limbs> unit(1, host::one(field));
            lincomb(field, qprime.data(), n, unit, fold_in.data());
            for (int j = 0; j < 4; ++j) lincomb(field, q.at((size_t)j * n * 32), n, unit, fold_in.at((size_t)(j + 1) * n * 32));
            std::vector<Limbs> p4(5);
            { Limbs acc = host::one(field); for (int j = 4; j >= 0; --j) { p4[j] = acc; acc = host::mul(field, acc, x4); } }
            lincomb(field, fold_in.data(), n, p4, p_poly.data());
            ms_multiopen = tm.stop();
            // the fold is linear: p(x3) == sum_j x4-power_j * (folded inputs)(x3)
            const std::vector<Limbs> parts = eval_polynomials(field, fold_in.data(), n, 5, x3);
            Limbs want{0, 0, 0, 0};
            for (int j = 0; j < 5; ++j) want = host::add(field, want, host::mul(field, p4[j], parts[j]));
            expect(eval_polynomials(field, p_poly.data(), n, 1, x3)[0] == want, "multiopen x4 fold evaluates consistently");
            // a division is exact up to the dropped remainder: q_0(X) == divided_0(X) (X - x) + q_0(x) at the point x3
            const Limbs d0 = eval_polynomials(field, divided.data(), n, 1, x3)[0], q0x = eval_polynomials(field, q.data(), n, 1, x)[0];
            expect(ev3[0] == host::add(field, host::mul(field, d0, host::sub(field, x3, x)), q0x), "kate_division identity at x3");
        }

        // keygen_vk / keygen_pk once per proving key: fixed + sigma columns (commit, iNTT, coset NTT), l0 / l_blind / l_last cosets
        double ms_keygen = 0;
        {
            SplitMix kr{0x6e9};
            const struct { int count; Kind kind; bool commit; } groups[3] = {{N_FIXED, witness ? Kind::Flag : Kind::Full, true}, {N_SIGMA, Kind::Full, true}, {3, witness ? Kind::Flag : Kind::Full, false}};
            for (const auto& g : groups) {
                for (int first = 0; first < g.count; first += (int)batch) {
                    const size_t b = std::min(batch, (size_t)(g.count - first));
                    for (size_t c = 0; c < b; ++c) fill_witness_column(g.kind, false, kr, word_bits, n, host_cols.data() + c * n);
                    cols.upload(host_cols.data(), b * n * 32);
                    if (g.kind != Kind::Full) check(trh_field_op_dev((int)field, 6, cols.data(), nullptr, cols.data(), b * n, nullptr), "to_mont");
                    for (size_t i = 0; i < b; ++i) blinds[i] = kr.element();
                    Timer tk;
                    if (g.commit) params.commit_lagrange_batch(cols, b, std::vector<Limbs>(blinds.begin(), blinds.begin() + b));
                    dom.lagrange_to_coeff(cols.data(), b);
                    dom.coeff_to_extended_blocks(cols.data(), ext.data(), b, D);
                    ms_keygen += tk.stop();
                }
            }
        }

        // IPA opening of the folded polynomial
        for (size_t i = 0; i < n; ++i) host_cols[i] = rng.element();
        s_poly.upload(host_cols.data(), n * 32);
        Transcript tr;
        trh_transcript_t tcb{&tr, tr_write_point, tr_write_scalar, tr_squeeze};
        SplitMix prng{0x99};
        Timer t8;
        const auto cfp = ipa_create_proof(params, p_poly, rng.element(), x_eval, s_poly, rng.element(), tcb, rng_scalar, &prng);
        ms_ipa = t8.stop();
        (void)cfp;
        expect(tr.points == 1 + 2 * (int)k && tr.scalars == 2, "IPA transcript: S, L_j / R_j per round, then c and f");

        // --overlap: the per-column phase once more over resident batches, step by step and with the transforms of batch i - 1 on a SECOND
        // context (own scratch, own stream, a second host thread: trh_ctx_create / trh_ctx_set_current) while this thread commits batch i
        double loop_seq_ms = 0, loop_ovl_ms = 0;
        if (overlap) {
            DeviceBuffer all_cols((size_t)lag_total * n * 32), work((size_t)lag_total * n * 32);
            SplitMix r2{0xc01};
            for (int done = 0; done < lag_total; done += (int)batch) {
                const size_t b = std::min(batch, (size_t)(lag_total - done));
                for (size_t c = 0; c < b; ++c) {
                    if (witness) fill_witness_column(kinds[done + c].first, kinds[done + c].second, r2, word_bits, n, host_cols.data() + c * n);
                    else for (size_t i = 0; i < n; ++i) host_cols[c * n + i] = r2.element();
                }
                all_cols.upload(host_cols.data(), b * n * 32, (size_t)done * n * 32);
                if (witness) check(trh_field_op_dev((int)field, 6, all_cols.at((size_t)done * n * 32), nullptr, all_cols.at((size_t)done * n * 32), b * n, nullptr), "to_mont");
            }
            check(trh_stream_synchronize(nullptr), "sync");
            trh_ctx_t ctx2 = nullptr;
            check(trh_ctx_create(0, &ctx2), "ctx_create");
            int bad = 0;
            auto transforms = [&](void* cols_dev, size_t b) {  // on the second context and its own stream
                if (trh_ctx_set_current(ctx2) != TRH_OK) { ++bad; return; }
                void* s2 = trh_ctx_stream(ctx2);  // the context's own stream: the null stream would serialise with the commitments
                try {
                    dom.lagrange_to_coeff(cols_dev, b, s2);
                    dom.coeff_to_extended_blocks(cols_dev, ext.data(), b, D, s2);
                    (void)eval_polynomials(field, cols_dev, n, b, x_eval, s2);
                    check(trh_stream_synchronize(s2), "sync");
                } catch (const std::exception&) { ++bad; }
                (void)trh_ctx_set_current(nullptr);
            };
            for (int pass = 0; pass < 2; ++pass) {
                for (int done = 0; done < lag_total; done += (int)batch) {  // fresh copies: the transforms work in place
                    const size_t b = std::min(batch, (size_t)(lag_total - done));
                    std::vector<Limbs> unit(1, host::one(field));
                    for (size_t c = 0; c < b; ++c) lincomb(field, all_cols.at(((size_t)done + c) * n * 32), n, unit, work.at(((size_t)done + c) * n * 32));
                }
                if (pass == 0) transforms(work.data(), 1);  // the second context builds its twiddle tables once (column 0 is restored below)
                if (pass == 0) { std::vector<Limbs> unit(1, host::one(field)); lincomb(field, all_cols.data(), n, unit, work.data()); }
                check(trh_stream_synchronize(nullptr), "sync");
                const double t0 = now_ms();
                void* prev = nullptr;
                size_t prev_b = 0;
                for (int done = 0; done < lag_total; done += (int)batch) {
                    const size_t b = std::min(batch, (size_t)(lag_total - done));
                    std::thread th;
                    if (prev && pass == 1) th = std::thread(transforms, prev, prev_b);
                    else if (prev) transforms(prev, prev_b);
                    for (size_t i = 0; i < b; ++i) blinds[i] = rng.element();
                    std::vector<Point> pts(b);
                    check(trh_commit_batch_dev(params.g_lagrange().handle(), work.at((size_t)done * n * 32), n, b, (const uint64_t*)blinds.data(), nullptr, (uint64_t*)pts.data()), "commit");
                    if (th.joinable()) th.join();
                    prev = work.at((size_t)done * n * 32);
                    prev_b = b;
                }
                transforms(prev, prev_b);
                check(trh_stream_synchronize(nullptr), "sync");
                (pass == 0 ? loop_seq_ms : loop_ovl_ms) = now_ms() - t0;
            }
            trh_ctx_destroy(ctx2);
            expect(bad == 0, "two-context column loop");
        }

        const double total = ms_lookup + ms_commit + ms_intt + ms_ext + ms_evals + ms_h + ms_commit_coeff + ms_ext_inv + ms_multiopen + ms_ipa;
        std::printf("{\"driver\": \"examples/replay.cpp\", \"word_bits\": %d, \"k\": %u, \"batch\": %zu, \"columns\": \"%s\", \"extended_domain\": \"5 of 8 coset blocks\", \"checks_failed\": %d, \"setup_ms\": %.3f, \"setup_tables_GB\": %.3f, \"keygen_ms\": %.3f, "
                    "\"ms\": {\"lookup_permute\": %.3f, \"commit_lagrange\": %.3f, \"lagrange_to_coeff\": %.3f, \"coeff_to_extended\": %.3f, \"evals\": %.3f, \"h_eval\": %.3f, \"commit\": %.3f, "
                    "\"extended_to_coeff\": %.3f, \"multiopen_folds\": %.3f, \"ipa\": %.3f}, \"ms_total\": %.3f, \"column_loop_ms\": {\"step_by_step\": %.3f, \"two_contexts_overlapped\": %.3f}, \"clock\": \"%s\"}\n",
                    word_bits, k, batch, witness ? "witness" : "random", failures, setup_ms, tables_gb, ms_keygen, ms_lookup, ms_commit, ms_intt, ms_ext, ms_evals, ms_h, ms_commit_coeff, ms_ext_inv, ms_multiopen, ms_ipa, total,
                    loop_seq_ms, loop_ovl_ms, g_host_clock ? "host clock between stream synchronisations" : "device events around every step (trh_event_*)");
        trh_shutdown();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    return failures ? 1 : 0;
}
