// trh.hpp -- C++17 host side over the C ABI of libtrh.so (include/trh.h), header-only.
//
// The reference is compiled Rust and the build image has no Rust toolchain, so the host layer a maintainer would
// write in the halo2_proofs fork is mirrored here in C++ with the reference's names, argument
// meaning and failure behaviour (the Rust functions panic on a violated precondition; these throw trh::Error):
//
//   trh::best_multiexp / trh::best_fft      halo2_proofs::arithmetic::{best_multiexp, best_fft}
//   trh::Params                             poly::commitment::Params  { commit, commit_lagrange } (+ device batch forms)
//   trh::EvaluationDomain                   poly::EvaluationDomain    { lagrange_to_coeff, coeff_to_extended,
//                                                                      extended_to_coeff, divide_by_vanishing_poly }
//   trh::Expression + compile_gates         plonk::Expression<F> and the y-folded evaluation of the gate polynomials
//   trh::ipa_create_proof                   poly::commitment::create_proof (IPA opening)
//
// Reference call sites of all of these: /root/reference/src/test_utils.rs:21-49, 89-104 (through keygen_* and
// create_proof of the halo2_proofs crate pinned at /root/reference/Cargo.lock:619-621).
// examples/replay.cpp is the native driver built on this header; tests/test_gpu_native.py runs it.
#ifndef TRH_HPP
#define TRH_HPP

#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "trh.h"

namespace trh {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void check(int rc, const char* what) {
    if (rc != TRH_OK) throw Error(std::string(what) + ": libtrh error " + std::to_string(rc) + ": " + trh_last_error());
}
inline void require(bool cond, const char* what) {  // the reference's assert! / assert_eq!
    if (!cond) throw Error(std::string("assertion failed: ") + what);
}

using Limbs = std::array<uint64_t, 4>;  // pasta_curves::Fp / Fq in memory: Montgomery form, little-endian limbs
struct Affine { Limbs x, y; };           // all-zero = identity (the Rust shim repacks pasta's flagged Affine)
struct Point { Limbs x, y, z; };         // pasta's Point layout, normalised to Z = 1 by libtrh
enum class Curve : int { Pallas = TRH_PALLAS, Vesta = TRH_VESTA };
enum class Field : int { Fp = TRH_FP, Fq = TRH_FQ };
inline Field scalar_field(Curve c) { return c == Curve::Pallas ? Field::Fq : Field::Fp; }

// ---- a little host field arithmetic (constants of expressions, checks in the example driver) ---------------
namespace host {
using u128 = unsigned __int128;
inline const Limbs& modulus(Field f) {
    static const Limbs P{0x992d30ed00000001ull, 0x224698fc094cf91bull, 0, 0x4000000000000000ull};
    static const Limbs Q{0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0, 0x4000000000000000ull};
    return f == Field::Fp ? P : Q;
}
inline bool geq(const Limbs& a, const Limbs& b) {
    for (int i = 3; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i];
    return true;
}
inline Limbs sub_raw(const Limbs& a, const Limbs& b) {
    Limbs r; u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - b[i] - br; r[i] = (uint64_t)d; br = (d >> 64) & 1; }
    return r;
}
inline Limbs add(Field f, const Limbs& a, const Limbs& b) {
    Limbs r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return geq(r, modulus(f)) ? sub_raw(r, modulus(f)) : r;  // a + b < 2m < 2^256
}
inline Limbs sub(Field f, const Limbs& a, const Limbs& b) {
    if (geq(a, b)) return sub_raw(a, b);
    Limbs t = sub_raw(modulus(f), b); return add(f, a, t);
}
inline Limbs neg(Field f, const Limbs& a) { return sub(f, Limbs{0, 0, 0, 0}, a); }
inline uint64_t neg_inv64(Field f) { return f == Field::Fp ? 0x992d30ecffffffffull : 0x8c46eb20ffffffffull; }  // -m^-1 mod 2^64 (m = 1 mod 2^32)
// Montgomery product a b / 2^256 mod m (CIOS)
inline Limbs mul(Field f, const Limbs& a, const Limbs& b) {
    const Limbs& m = modulus(f);
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)t[j] + (u128)a[i] * b[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t q = t[0] * neg_inv64(f);
        c = (u128)t[0] + (u128)q * m[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)t[j] + (u128)q * m[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64); t[5] = 0;
    }
    Limbs r{t[0], t[1], t[2], t[3]};
    return (t[4] || geq(r, m)) ? sub_raw(r, m) : r;
}
inline Limbs one(Field f) {  // R mod m = 2^256 - m * floor(2^256 / m) = 2^256 - 3m for these moduli (m just above 2^254)
    const Limbs& m = modulus(f);
    Limbs z{0, 0, 0, 0}, r = sub_raw(z, m); r = sub_raw(r, m); r = sub_raw(r, m);
    return r;
}
// a^(m - 2) (Fermat); inv(0) = 0
inline Limbs inv(Field f, const Limbs& a) {
    Limbs e = modulus(f);
    e[0] -= 2;  // the low limbs of both moduli are far above 2: no borrow
    Limbs r = one(f);
    for (int i = 255; i >= 0; --i) {
        r = mul(f, r, r);
        if ((e[i / 64] >> (i % 64)) & 1) r = mul(f, r, a);
    }
    return r;
}
inline Limbs from_u64(Field f, uint64_t v) {  // v * R: double-and-add on the Montgomery one
    Limbs acc{0, 0, 0, 0}, base = one(f);
    for (int i = 0; i < 64; ++i) { if ((v >> i) & 1) acc = add(f, acc, base); base = add(f, base, base); }
    return acc;
}
}  // namespace host

// ---- halo2_proofs::arithmetic -------------------------------------------------------------------------------
inline Point best_multiexp(Curve c, const std::vector<Limbs>& coeffs, const std::vector<Affine>& bases) {
    require(coeffs.size() == bases.size(), "coeffs.len() == bases.len()");
    Point out;
    const auto fn = c == Curve::Pallas ? trh_best_multiexp_pallas : trh_best_multiexp_vesta;
    check(fn((const uint64_t*)coeffs.data(), (const uint64_t*)bases.data(), coeffs.size(), (uint64_t*)&out), "best_multiexp");
    return out;
}
inline void best_fft(Field f, std::vector<Limbs>& a, const Limbs& omega, uint32_t log_n) {
    require(a.size() == (size_t)1 << log_n, "a.len() == 1 << log_n");
    const auto fn = f == Field::Fp ? trh_best_fft_fp : trh_best_fft_fq;
    check(fn((uint64_t*)a.data(), omega.data(), log_n), "best_fft");
}

// the same over a batch of host slices with one omega (pipelined over PCIe)
inline void best_fft_batch(Field f, std::vector<std::vector<Limbs>*>& columns, const Limbs& omega, uint32_t log_n) {
    std::vector<uint64_t*> ptrs;
    for (auto* c : columns) { require(c->size() == (size_t)1 << log_n, "a.len() == 1 << log_n"); ptrs.push_back((uint64_t*)c->data()); }
    check((f == Field::Fp ? trh_best_fft_batch_fp : trh_best_fft_batch_fq)(ptrs.data(), ptrs.size(), omega.data(), log_n), "best_fft_batch");
}

// ---- device memory -------------------------------------------------------------------------------------------
class DeviceBuffer {
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes) : bytes_(bytes) { check(trh_malloc(&p_, bytes ? bytes : 16), "trh_malloc"); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept { std::swap(p_, o.p_); std::swap(bytes_, o.bytes_); return *this; }
    ~DeviceBuffer() { if (p_) (void)trh_free(p_); }
    void* data() const { return p_; }
    void* at(size_t byte_offset) const { return (char*)p_ + byte_offset; }
    size_t size() const { return bytes_; }
    void upload(const void* host, size_t bytes, size_t offset = 0) { require(offset + bytes <= bytes_, "upload range"); check(trh_memcpy_h2d(at(offset), host, bytes), "h2d"); }
    void download(void* host, size_t bytes, size_t offset = 0) const { require(offset + bytes <= bytes_, "download range"); check(trh_memcpy_d2h(host, at(offset), bytes), "d2h"); }
private:
    void* p_ = nullptr;
    size_t bytes_ = 0;
};

// ---- poly::commitment::Params ---------------------------------------------------------------------------------
class Bases {
public:
    Bases() = default;
    Bases(Curve c, const std::vector<Affine>& xy) {
        check((c == Curve::Pallas ? trh_bases_create_pallas : trh_bases_create_vesta)((const uint64_t*)xy.data(), xy.size(), &h_), "bases_create");
    }
    static Bases generate(Curve c, uint64_t s0, uint64_t d, size_t n) { Bases b; check(trh_bases_generate((int)c, s0, d, 0, n, &b.h_), "bases_generate"); return b; }
    Bases(const Bases&) = delete;
    Bases& operator=(const Bases&) = delete;
    Bases(Bases&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Bases& operator=(Bases&& o) noexcept { std::swap(h_, o.h_); return *this; }
    ~Bases() { if (h_) trh_bases_destroy(h_); }
    trh_bases_t handle() const { return h_; }
    size_t len() const { return trh_bases_len(h_); }
    int precompute(int window_bits = 0) { check(trh_bases_precompute(h_, window_bits), "bases_precompute"); return trh_bases_precomputed_window_bits(h_); }
    // setup-time: the calling context's MSM scratch for batches of `batch` MSMs over the first n bases (the first batch then allocates nothing)
    void reserve(size_t n, size_t batch) const { check(trh_bases_reserve(h_, n, batch), "bases_reserve"); }
    std::vector<Affine> download() const { std::vector<Affine> v(len()); check(trh_bases_download(h_, 0, v.size(), (uint64_t*)v.data()), "bases_download"); return v; }
    Point msm(const std::vector<Limbs>& scalars, size_t offset = 0) const {
        Point out; check(trh_msm(h_, offset, (const uint64_t*)scalars.data(), scalars.size(), 1, (uint64_t*)&out), "msm"); return out;
    }
private:
    trh_bases_t h_ = nullptr;
};

class Params {
public:
    // g, g_lagrange: n = 2^k points each; w, u as in commitment::Params.  Both base sets are kept resident with w appended.
    Params(Curve c, uint32_t k, std::vector<Affine> g, std::vector<Affine> g_lagrange, const Affine& w, const Affine& u, bool fixed_base_tables = true)
        : curve(c), k(k), n((size_t)1 << k), w(w), u(u) {
        require(g.size() == n && g_lagrange.size() == n, "g.len() == g_lagrange.len() == 1 << k");
        g.push_back(w); g_lagrange.push_back(w);
        g_ = Bases(c, g); g_lagrange_ = Bases(c, g_lagrange);
        if (fixed_base_tables) precompute();
    }
    // synthetic resident generators (the example driver / benchmarks): (s0 + i d) G
    Params(Curve c, uint32_t k, uint64_t s0, uint64_t d, bool fixed_base_tables = true) : curve(c), k(k), n((size_t)1 << k) {
        g_ = Bases::generate(c, s0, d, n + 1); g_lagrange_ = Bases::generate(c, s0 + 77, d + 2, n + 1);
        const auto gh = g_.download(); w = gh[n]; u = Bases::generate(c, 4242, 1, 1).download()[0];
        if (fixed_base_tables) precompute();
    }
    void precompute() {
        for (Bases* b : {&g_, &g_lagrange_}) { try { b->precompute(0); } catch (const Error&) { /* outside the table range: per-window path */ } }
        // the opening's base set g || w || u with its own table: every MSM of trh_ipa_create_proof then runs in fixed-base mode
        try {
            std::vector<Affine> gwu = g_.download();
            gwu.push_back(u);
            Bases b(curve, gwu);
            b.precompute(0);
            ipa_ = std::move(b);
        } catch (const Error&) { /* the opening uses g || w and the per-window path */ }
    }
    Point commit(const std::vector<Limbs>& poly, const Limbs& blind) const { return commit_host(g_, poly, blind); }
    Point commit_lagrange(const std::vector<Limbs>& poly, const Limbs& blind) const { return commit_host(g_lagrange_, poly, blind); }
    // `batch` polynomials of n coefficients back to back in device memory
    std::vector<Point> commit_batch(const DeviceBuffer& polys, size_t batch, const std::vector<Limbs>& blinds, void* stream = nullptr) const { return commit_dev(g_, polys, batch, blinds, stream); }
    std::vector<Point> commit_lagrange_batch(const DeviceBuffer& polys, size_t batch, const std::vector<Limbs>& blinds, void* stream = nullptr) const { return commit_dev(g_lagrange_, polys, batch, blinds, stream); }
    // the same for polynomials in HOST memory, one vector per column (trh_commit_batch_host: chunked uploads under the MSMs)
    std::vector<Point> commit_batch_host(const std::vector<const std::vector<Limbs>*>& polys, const std::vector<Limbs>& blinds) const { return commit_hosts(g_, polys, blinds); }
    std::vector<Point> commit_lagrange_batch_host(const std::vector<const std::vector<Limbs>*>& polys, const std::vector<Limbs>& blinds) const { return commit_hosts(g_lagrange_, polys, blinds); }
    const Bases& g() const { return g_; }
    const Bases& g_lagrange() const { return g_lagrange_; }
    const Bases& ipa_bases() const { return ipa_.handle() ? ipa_ : g_; }  // g || w || u with tables when precompute() built it
    // setup-time sizing of the calling context's scratch for commit batches of `batch` columns and for the opening's round MSMs
    void reserve(size_t batch) const {
        g_.reserve(n + 1, batch);
        g_lagrange_.reserve(n + 1, batch);
        if (ipa_.handle()) ipa_.reserve(n + 2, 2);
    }

    Curve curve;
    uint32_t k;
    size_t n;
    Affine w{}, u{};
private:
    Point commit_host(const Bases& b, const std::vector<Limbs>& poly, const Limbs& blind) const {
        require(poly.size() == n, "poly.len() == params.n");
        std::vector<Limbs> sc(poly); sc.push_back(blind);
        return b.msm(sc);
    }
    std::vector<Point> commit_hosts(const Bases& b, const std::vector<const std::vector<Limbs>*>& polys, const std::vector<Limbs>& blinds) const {
        require(blinds.size() == polys.size(), "one blind per polynomial");
        std::vector<const uint64_t*> ptrs;
        for (const auto* p : polys) { require(p->size() == n, "poly.len() == params.n"); ptrs.push_back((const uint64_t*)p->data()); }
        std::vector<Point> out(polys.size());
        check(trh_commit_batch_host(b.handle(), ptrs.data(), n, ptrs.size(), (const uint64_t*)blinds.data(), (uint64_t*)out.data()), "commit_batch_host");
        return out;
    }
    std::vector<Point> commit_dev(const Bases& b, const DeviceBuffer& polys, size_t batch, const std::vector<Limbs>& blinds, void* stream) const {
        require(blinds.size() == batch && polys.size() >= batch * n * 32, "batch x n coefficients and one blind per polynomial");
        std::vector<Point> out(batch);
        check(trh_commit_batch_dev(b.handle(), polys.data(), n, batch, (const uint64_t*)blinds.data(), stream, (uint64_t*)out.data()), "commit_batch");
        return out;
    }
    Bases g_, g_lagrange_, ipa_;
};

// ---- poly::EvaluationDomain -----------------------------------------------------------------------------------
class EvaluationDomain {
public:
    EvaluationDomain(Field f, uint32_t j, uint32_t k) : field(f), k(k), n((size_t)1 << k) { check(trh_domain_create((int)f, j, k, &d_), "EvaluationDomain::new"); extended_k = trh_domain_extended_k(d_); }
    EvaluationDomain(const EvaluationDomain&) = delete;
    EvaluationDomain& operator=(const EvaluationDomain&) = delete;
    ~EvaluationDomain() { if (d_) trh_domain_destroy(d_); }
    size_t extended_len() const { return (size_t)1 << extended_k; }
    // setup-time: tables and scratch of the per-column transforms for batches of `batch` polynomials on the calling context
    void reserve(size_t batch) const { check(trh_domain_reserve(d_, batch), "domain_reserve"); }
    Limbs constant(int which) const { Limbs v; check(trh_domain_constant(d_, which, v.data()), "domain_constant"); return v; }
    Limbs get_omega() const { return constant(0); }
    Limbs get_extended_omega() const { return constant(2); }
    // batch x 2^k (or 2^extended_k) elements, back to back in device memory
    void lagrange_to_coeff(void* a_dev, size_t batch, void* stream = nullptr) const { check(trh_domain_lagrange_to_coeff(d_, a_dev, batch, stream), "lagrange_to_coeff"); }
    void coeff_to_extended(const void* coeff_dev, void* ext_dev, size_t batch, void* stream = nullptr) const { check(trh_domain_coeff_to_extended(d_, coeff_dev, ext_dev, batch, stream), "coeff_to_extended"); }
    void extended_to_coeff(void* a_dev, size_t batch, void* stream = nullptr) const { check(trh_domain_extended_to_coeff(d_, a_dev, batch, stream), "extended_to_coeff"); }
    void divide_by_vanishing_poly(void* a_dev, size_t batch, void* stream = nullptr) const { check(trh_domain_divide_by_vanishing_poly(d_, a_dev, batch, stream), "divide_by_vanishing_poly"); }

    // the extended domain as coset blocks (trh.h): n_blocks x 2^k values per polynomial, block r entry q = coeff_to_extended's entry
    // q * 2^(extended_k - k) + r; the quotient needs quotient_blocks() = j - 1 of them
    uint32_t quotient_blocks() const { return trh_domain_quotient_blocks(d_); }
    void coeff_to_extended_blocks(const void* coeff_dev, void* ext_dev, size_t batch, uint32_t n_blocks, void* stream = nullptr) const {
        check(trh_domain_coeff_to_extended_blocks(d_, coeff_dev, ext_dev, batch, n_blocks, stream), "coeff_to_extended_blocks");
    }
    void blocks_to_quotient(void* num_blocks_dev, void* h_coeff_dev, bool divide_by_vanishing = true, void* stream = nullptr) const {
        check(trh_domain_blocks_to_quotient(d_, num_blocks_dev, h_coeff_dev, divide_by_vanishing ? 1 : 0, stream), "blocks_to_quotient");
    }
    // host polynomials, one vector per column (pipelined over PCIe): lagrange_to_coeff in place; coeff_to_extended 2^k -> 2^extended_k;
    // extended_to_coeff in place on one polynomial, optionally preceded by divide_by_vanishing_poly (the caller truncates)
    void lagrange_to_coeff_host(std::vector<std::vector<Limbs>*>& cols) const {
        std::vector<uint64_t*> ptrs;
        for (auto* c : cols) { require(c->size() == n, "a.len() == 1 << k"); ptrs.push_back((uint64_t*)c->data()); }
        check(trh_domain_lagrange_to_coeff_host(d_, ptrs.data(), ptrs.size()), "lagrange_to_coeff_host");
    }
    void coeff_to_extended_host(const std::vector<const std::vector<Limbs>*>& coeff, std::vector<std::vector<Limbs>*>& ext) const {
        require(coeff.size() == ext.size(), "one output per input");
        std::vector<const uint64_t*> in;
        std::vector<uint64_t*> out;
        for (const auto* c : coeff) { require(c->size() == n, "a.len() == 1 << k"); in.push_back((const uint64_t*)c->data()); }
        for (auto* e : ext) { require(e->size() == extended_len(), "ext.len() == 1 << extended_k"); out.push_back((uint64_t*)e->data()); }
        check(trh_domain_coeff_to_extended_host(d_, in.data(), out.data(), in.size()), "coeff_to_extended_host");
    }
    void extended_to_coeff_host(std::vector<Limbs>& a, bool divide_by_vanishing_first = false) const {
        require(a.size() == extended_len(), "a.len() == 1 << extended_k");
        check(trh_domain_extended_to_coeff_host(d_, (uint64_t*)a.data(), divide_by_vanishing_first ? 1 : 0), "extended_to_coeff_host");
    }

    Field field;
    uint32_t k, extended_k = 0;
    size_t n;
private:
    trh_domain_t d_ = nullptr;
};

// ---- plonk::Expression<F> --------------------------------------------------------------------------------------
struct Expression;
using Expr = std::shared_ptr<const Expression>;
struct Expression {
    enum Kind { Constant, Selector, Fixed, Advice, Instance, Negated, Sum, Product, Scaled } kind;
    Limbs value{};      // Constant, Scaled
    uint32_t column = 0;
    int32_t rotation = 0;
    Expr a, b;
    int degree() const {
        switch (kind) {
            case Constant: return 0;
            case Selector: case Fixed: case Advice: case Instance: return 1;
            case Negated: case Scaled: return a->degree();
            case Sum: return std::max(a->degree(), b->degree());
            default: return a->degree() + b->degree();
        }
    }
};
inline Expr constant(const Limbs& v) { auto e = std::make_shared<Expression>(); e->kind = Expression::Constant; e->value = v; return e; }
inline Expr query(Expression::Kind k, uint32_t column, int32_t rotation = 0) { auto e = std::make_shared<Expression>(); e->kind = k; e->column = column; e->rotation = rotation; return e; }
inline Expr advice(uint32_t c, int32_t r = 0) { return query(Expression::Advice, c, r); }
inline Expr fixed(uint32_t c, int32_t r = 0) { return query(Expression::Fixed, c, r); }
inline Expr instance(uint32_t c, int32_t r = 0) { return query(Expression::Instance, c, r); }
inline Expr selector(uint32_t c) { return query(Expression::Selector, c, 0); }
inline Expr node(Expression::Kind k, Expr a, Expr b = nullptr) { auto e = std::make_shared<Expression>(); e->kind = k; e->a = std::move(a); e->b = std::move(b); return e; }
inline Expr operator-(const Expr& a) { return node(Expression::Negated, a); }
inline Expr operator+(const Expr& a, const Expr& b) { return node(Expression::Sum, a, b); }
inline Expr operator-(const Expr& a, const Expr& b) { return node(Expression::Sum, a, -b); }  // halo2: a - b == a + (-b)
inline Expr operator*(const Expr& a, const Expr& b) { return node(Expression::Product, a, b); }
inline Expr scaled(const Expr& a, const Limbs& v) { auto e = std::make_shared<Expression>(); e->kind = Expression::Scaled; e->a = a; e->value = v; return e; }

struct Program {
    Field field;
    std::vector<trh_expr_insn_t> insns;
    std::vector<Limbs> consts;                                  // index 0: the folding challenge y
    std::vector<std::pair<Expression::Kind, uint32_t>> columns;  // resident column of every slot, in slot order
    int max_degree = 0;
};

namespace detail {
inline int need(const Expr& e) {  // Sethi-Ullman number: stack entries the sub-expression needs
    switch (e->kind) {
        case Expression::Negated: case Expression::Scaled: return need(e->a);
        case Expression::Sum: case Expression::Product: { const int x = need(e->a), y = need(e->b); return x != y ? std::max(x, y) : x + 1; }
        default: return 1;
    }
}
struct Lowering {
    Program& p;
    uint32_t constant(const Limbs& v) {
        for (size_t i = 1; i < p.consts.size(); ++i) if (p.consts[i] == v) return (uint32_t)i;
        p.consts.push_back(v); return (uint32_t)p.consts.size() - 1;
    }
    uint32_t column(Expression::Kind k, uint32_t c) {
        for (size_t i = 0; i < p.columns.size(); ++i) if (p.columns[i].first == k && p.columns[i].second == c) return (uint32_t)i;
        p.columns.emplace_back(k, c); return (uint32_t)p.columns.size() - 1;
    }
    void op(uint32_t o, uint32_t a = 0, int32_t r = 0) { p.insns.push_back(trh_expr_insn_t{o, a, r}); }
    void emit(const Expr& e) {
        switch (e->kind) {
            case Expression::Constant: op(TRH_EXPR_PUSH_CONST, constant(e->value)); break;
            case Expression::Selector: case Expression::Fixed: case Expression::Advice: case Expression::Instance:
                op(TRH_EXPR_PUSH_COLUMN, column(e->kind, e->column), e->rotation); break;
            case Expression::Negated: emit(e->a); op(TRH_EXPR_NEG); break;
            case Expression::Scaled: emit(e->a); op(TRH_EXPR_MUL_CONST, constant(e->value)); break;
            case Expression::Sum:
                if (e->b->kind == Expression::Negated) {  // a + (-b): one SUB
                    const Expr &x = e->a, &y = e->b->a;
                    if (need(y) > need(x)) { emit(y); emit(x); op(TRH_EXPR_SUB); op(TRH_EXPR_NEG); }
                    else { emit(x); emit(y); op(TRH_EXPR_SUB); }
                } else {
                    const bool swap = need(e->b) > need(e->a);
                    emit(swap ? e->b : e->a); emit(swap ? e->a : e->b); op(TRH_EXPR_ADD);
                }
                break;
            case Expression::Product:
                if (e->a == e->b) { emit(e->a); op(TRH_EXPR_SQR); }
                else { const bool swap = need(e->b) > need(e->a); emit(swap ? e->b : e->a); emit(swap ? e->a : e->b); op(TRH_EXPR_MUL); }
                break;
        }
    }
};
}  // namespace detail

// gate polynomials folded into one output with the challenge y: h = h * y + gate (the order create_proof uses)
inline Program compile_gates(Field f, const std::vector<Expr>& gates, const Limbs& y) {
    Program p; p.field = f; p.consts.push_back(y);
    detail::Lowering lo{p};
    for (const Expr& g : gates) { p.max_degree = std::max(p.max_degree, g->degree()); lo.emit(g); lo.op(TRH_EXPR_FOLD, 0); }
    lo.op(TRH_EXPR_STORE_ACC, 0);
    return p;
}

// one output column per expression (numerator / denominator of a grand-product argument)
inline Program compile_outputs(Field f, const std::vector<Expr>& exprs) {
    Program p; p.field = f;
    p.consts.push_back(Limbs{0, 0, 0, 0});  // slot 0 is reserved for a challenge in compile_gates; unused here
    detail::Lowering lo{p};
    uint32_t i = 0;
    for (const Expr& e : exprs) { p.max_degree = std::max(p.max_degree, e->degree()); lo.emit(e); lo.op(TRH_EXPR_STORE_TOP, i++); }
    return p;
}

class GateEvaluator {
public:
    explicit GateEvaluator(Program p, size_t n_outputs = 1) : program(std::move(p)), n_outputs(n_outputs) {
        check(trh_expr_create((int)program.field, program.insns.data(), program.insns.size(), (const uint64_t*)program.consts.data(), program.consts.size(),
                              program.columns.size(), n_outputs, 0, &e_), "expr_create");
    }
    GateEvaluator(const GateEvaluator&) = delete;
    GateEvaluator& operator=(const GateEvaluator&) = delete;
    ~GateEvaluator() { if (e_) trh_expr_destroy(e_); }
    void set_challenge(const Limbs& y) { check(trh_expr_set_const(e_, 0, y.data()), "expr_set_const"); }
    // columns[i]: device pointer of program.columns[i] (2^log_n elements); out_dev: 2^log_n elements
    void eval(const std::vector<const void*>& columns, void* out_dev, uint32_t log_n, uint32_t rot_step, void* stream = nullptr) const {
        require(columns.size() == program.columns.size(), "one device column per program column");
        void* outs[1] = {out_dev};
        check(trh_expr_eval_dev(e_, columns.data(), outs, log_n, rot_step, stream), "expr_eval");
    }
    // columns in the coset-block layout: n_blocks x 2^block_log rows each, Rotation(r) stays inside its block
    void eval_blocks(const std::vector<const void*>& columns, void* out_dev, uint32_t block_log, uint32_t n_blocks, void* stream = nullptr) const {
        require(columns.size() == program.columns.size(), "one device column per program column");
        void* outs[1] = {out_dev};
        check(trh_expr_eval_blocks_dev(e_, columns.data(), outs, block_log, n_blocks, stream), "expr_eval_blocks");
    }
    void eval_outputs(const std::vector<const void*>& columns, const std::vector<void*>& outs, uint32_t log_n, uint32_t rot_step, void* stream = nullptr) const {
        require(columns.size() == program.columns.size() && outs.size() == n_outputs, "one device pointer per program column / output");
        check(trh_expr_eval_dev(e_, columns.data(), outs.data(), log_n, rot_step, stream), "expr_eval");
    }
    Program program;
    size_t n_outputs;
private:
    trh_expr_t e_ = nullptr;
};

// ---- poly::commitment::create_proof (IPA opening) -------------------------------------------------------------------
// transcript / rng are the caller's (BLAKE2b transcript and OsRng in the reference): plain C callbacks as in trh.h
inline std::pair<Limbs, Limbs> ipa_create_proof(const Params& params, const DeviceBuffer& p_poly, const Limbs& p_blind, const Limbs& x3, const DeviceBuffer& s_poly,
                                                const Limbs& s_blind, const trh_transcript_t& transcript, trh_rng_scalar_fn rng, void* rng_ctx, void* stream = nullptr) {
    require(p_poly.size() >= params.n * 32 && s_poly.size() >= params.n * 32, "px.len() == params.n");
    Limbs c, f;
    check(trh_ipa_create_proof(params.ipa_bases().handle(), (const uint64_t*)&params.u, params.k, p_poly.data(), p_blind.data(), x3.data(), s_poly.data(), s_blind.data(), &transcript, rng,
                               rng_ctx, stream, c.data(), f.data()), "ipa create_proof");
    return {c, f};
}

// ---- the remaining polynomial steps of create_proof on resident data -------------------------------------------------
// arithmetic::eval_polynomial for `batch` coefficient forms (n each, back to back) at one point
inline std::vector<Limbs> eval_polynomials(Field f, const void* polys_dev, size_t n, size_t batch, const Limbs& point, void* stream = nullptr) {
    std::vector<Limbs> out(batch);
    check(trh_poly_eval_batch_dev((int)f, polys_dev, n, batch, point.data(), stream, (uint64_t*)out.data()), "eval_polynomial");
    return out;
}
// poly/multiopen/prover.rs: sum_b coeffs[b] * polys[b]
inline void lincomb(Field f, const void* polys_dev, size_t n, const std::vector<Limbs>& coeffs, void* out_dev, void* stream = nullptr) {
    check(trh_poly_lincomb_dev((int)f, polys_dev, n, coeffs.size(), (const uint64_t*)coeffs.data(), out_dev, stream), "lincomb");
}
// plonk/lookup/prover.rs permute_expression_pair over the first usable_rows rows; throws where the Rust code returns
// Error::ConstraintSystemFailure (an input value that is not in the table)
inline void lookup_permute(Field f, const void* input_dev, const void* table_dev, size_t usable_rows, void* permuted_input_dev, void* permuted_table_dev, void* stream = nullptr) {
    check(trh_lookup_permute_dev((int)f, input_dev, table_dev, usable_rows, permuted_input_dev, permuted_table_dev, stream), "permute_expression_pair");
}

// all lookups of a proof at once: column l of inputs / tables / outputs at element offset l * row_stride
inline void lookup_permute_batch(Field f, const void* inputs_dev, const void* tables_dev, size_t usable_rows, size_t row_stride, size_t batch, void* permuted_inputs_dev,
                                 void* permuted_tables_dev, void* stream = nullptr) {
    check(trh_lookup_permute_batch_dev((int)f, inputs_dev, tables_dev, usable_rows, row_stride, batch, permuted_inputs_dev, permuted_tables_dev, stream), "permute_expression_pair (batch)");
}

// z[0] = z0, z[i + 1] = z[i] * num(row i) / den(row i): the product columns of the permutation argument
// (plonk/permutation/prover.rs) and of the lookup argument (plonk/lookup/prover.rs commit_product)
class GrandProduct {
public:
    GrandProduct(Field f, uint32_t k, const Expr& num, const Expr& den) : field(f), k(k), n((size_t)1 << k), ev_(compile_outputs(f, {num, den}), 2), num_(n * 32), den_(n * 32), ratio_(n * 32) {}
    const std::vector<std::pair<Expression::Kind, uint32_t>>& columns() const { return ev_.program.columns; }
    // columns[i]: device pointer of columns()[i]; z_dev: 2^k elements
    void compute(const std::vector<const void*>& cols, void* z_dev, uint32_t rot_step = 1, void* stream = nullptr) {
        ev_.eval_outputs(cols, {num_.data(), den_.data()}, k, rot_step, stream);
        check(trh_field_batch_invert_dev((int)field, den_.data(), n, stream), "batch_invert");
        check(trh_field_op_dev((int)field, 2 /* mul */, num_.data(), den_.data(), ratio_.data(), n, stream), "ratio");
        check(trh_field_prefix_product_dev((int)field, ratio_.data(), z_dev, n, stream), "prefix_product");
    }
    Field field;
    uint32_t k;
    size_t n;
private:
    GateEvaluator ev_;
    DeviceBuffer num_, den_, ratio_;
};

// Every product column of a proof at once, fixed-function (trh_product_terms_dev): a row is the list of the factors
// (x[i] + c y[i] + g) of one numerator or denominator product.  permutation chunk: numerator {v_j, omega^i column, beta delta^col_j, gamma},
// denominator {v_j, sigma_j, beta, gamma}; lookup: numerator {A, -, -, beta}{S, -, -, gamma}, denominator {A', -, -, beta}{S', -, -, gamma}
inline trh_product_term_t product_term(const void* x, const void* y, const Limbs& c, const Limbs& g) {
    trh_product_term_t t{};
    t.x = x; t.y = y;
    for (int i = 0; i < 4; ++i) { t.c[i] = c[i]; t.g[i] = g[i]; }
    return t;
}
// z_dev: rows x 2^k elements, z[r][0] = 1, z[r][i + 1] = z[r][i] * num_r(i) / den_r(i) (a zero denominator gives a zero ratio, as
// ff::BatchInvert followed by the multiplication does)
inline void grand_products(Field f, uint32_t k, const std::vector<std::vector<trh_product_term_t>>& num_rows, const std::vector<std::vector<trh_product_term_t>>& den_rows,
                           void* z_dev, void* stream = nullptr) {
    if (num_rows.size() != den_rows.size()) throw Error("grand_products: numerator / denominator row counts differ");
    const size_t rows = num_rows.size(), n = (size_t)1 << k;
    if (!rows) return;
    std::vector<trh_product_term_t> flat;
    std::vector<uint32_t> start{0};
    for (const auto* part : {&num_rows, &den_rows})
        for (const auto& row : *part) { flat.insert(flat.end(), row.begin(), row.end()); start.push_back((uint32_t)flat.size()); }
    DeviceBuffer nd(2 * rows * n * 32);
    check(trh_product_terms_dev((int)f, flat.data(), start.data(), (uint32_t)(2 * rows), n, nd.data(), stream), "product_terms");
    char* den = (char*)nd.data() + rows * n * 32;
    check(trh_field_batch_invert_mul_dev((int)f, den, nd.data(), rows * n, stream), "batch_invert_mul");
    check(trh_field_prefix_product_rows_dev((int)f, den, z_dev, n, rows, stream), "prefix_product_rows");
    check(trh_stream_synchronize(stream), "grand_products sync");  // nd is released on return
}

// arithmetic::kate_division by (X - z) for polynomials of n coefficients.  The powers of z and 1 / z are built once, on the stream
// the divisions will run on (the constructor takes it: tables built on another stream could still be in flight when divide() reads
// them); z = 0 (z_inv ignored) divides by X: the quotient is the coefficient list shifted down by one.
class KateDivider {
public:
    KateDivider(Field f, size_t n, const Limbs& z, const Limbs& z_inv, void* stream = nullptr)
        : field(f), n(n), zero_(z == Limbs{0, 0, 0, 0}), stream_(stream), pz_(zero_ ? 32 : n * 32), pzinv_(zero_ ? 32 : n * 32), scratch_(zero_ ? 32 : 2 * n * 32) {
        if (zero_) return;
        check(trh_field_powers_dev((int)f, pz_.data(), n, z.data(), stream), "powers");
        check(trh_field_powers_dev((int)f, pzinv_.data(), n, z_inv.data(), stream), "powers");
    }
    // a_dev: len coefficients (default: n; a shorter polynomial uses the prefixes of the power tables -- the multiopen sets divide by the
    // same point at lengths n, n - 1, ...), q_dev: len - 1 coefficients; runs on the constructor's stream
    void divide(const void* a_dev, void* q_dev, size_t len = 0) {
        if (len == 0) len = n;
        require(len <= n, "kate_division: polynomial longer than the divider's tables");
        if (zero_) {  // a(X) / X, remainder a_0 dropped: host-ordered device-to-device copy through the ABI's helpers
            std::vector<Limbs> tmp(len);
            check(trh_stream_synchronize(stream_), "sync");
            check(trh_memcpy_d2h(tmp.data(), a_dev, len * 32), "d2h");
            check(trh_memcpy_h2d(q_dev, tmp.data() + 1, (len - 1) * 32), "h2d");
            return;
        }
        check(trh_poly_kate_division_dev((int)field, a_dev, len, pz_.data(), pzinv_.data(), scratch_.data(), q_dev, stream_), "kate_division");
    }
    Field field;
    size_t n;
private:
    bool zero_;
    void* stream_;
    DeviceBuffer pz_, pzinv_, scratch_;
};

inline void init(int device = 0) { check(trh_init(device), "trh_init"); }

// An independent libtrh context on one device (own scratch, own lock, own stream): one per host thread that should overlap with others, one
// per GPU of a node.  bind() makes it the calling thread's context until unbind(); objects created while it is bound (Bases, Params,
// EvaluationDomain, DeviceBuffer) live on its device.
class Context {
public:
    explicit Context(int device = 0) { check(trh_ctx_create(device, &c_), "trh_ctx_create"); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ~Context() { if (c_) trh_ctx_destroy(c_); }
    void bind() const { check(trh_ctx_set_current(c_), "trh_ctx_set_current"); }
    static void unbind() { check(trh_ctx_set_current(nullptr), "trh_ctx_set_current"); }
    void* stream() const { return trh_ctx_stream(c_); }
    int device() const { return trh_ctx_device(c_); }
private:
    trh_ctx_t c_ = nullptr;
};

}  // namespace trh
#endif  // TRH_HPP
