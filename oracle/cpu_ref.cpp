// ORACLE (test infrastructure, NOT product code).
//
// CPU restatement of the halo2_proofs 0.2.0 hot path the reference drives through
// create_proof (reference call sites: /root/reference/src/test_utils.rs:21, 23-25, 41-49;
// crate pins /root/reference/Cargo.lock:619-621 (halo2_proofs fork, rev a95945254...),
// :847-858 (pasta_curves 0.4.1)).  The crates themselves are NOT vendored and there is no
// Rust toolchain here, so this follows the published algorithms:
//   * pasta_curves Fp/Fq: 4 x u64 little-endian limbs, Montgomery R = 2^256, fully reduced
//   * pallas/vesta: y^2 = x^3 + 5, Jacobian (X, Y, Z), identity Z = 0
//   * arithmetic::best_multiexp / multiexp_serial: chunk-per-thread Pippenger, window
//     c = 1 (n<4) / 3 (n<32) / ceil(ln n), segments = 256/c + 1, buckets None|Affine|Projective,
//     running-sum reduction
//   * arithmetic::best_fft: bit-reverse, sequential twiddle scan, radix-2 DIT butterflies
//
// PARITY UNPINNED BY THE REFERENCE (it holds no MSM/NTT/field golden vectors); pinned instead
// against oracle/pasta.py (Python big-int, published pasta constants) in tests/test_oracle.py.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// Build: g++ -O3 -march=native -std=c++17 -shared -fPIC -pthread cpu_ref.cpp -o libtrh_oracle.so
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

typedef uint64_t u64;
typedef unsigned __int128 u128;

namespace {

struct FpP {
    static constexpr u64 M[4] = {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0ULL, 0x4000000000000000ULL};
    static constexpr u64 INV = 0x992d30ecffffffffULL;
    static constexpr u64 R[4] = {0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL};
    static constexpr u64 R2[4] = {0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL};
};
struct FqP {
    static constexpr u64 M[4] = {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0ULL, 0x4000000000000000ULL};
    static constexpr u64 INV = 0x8c46eb20ffffffffULL;
    static constexpr u64 R[4] = {0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL};
    static constexpr u64 R2[4] = {0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL};
};
constexpr u64 FpP::M[4]; constexpr u64 FpP::R[4]; constexpr u64 FpP::R2[4];
constexpr u64 FqP::M[4]; constexpr u64 FqP::R[4]; constexpr u64 FqP::R2[4];

// ---- field element, Montgomery form --------------------------------------------------
template <class P>
struct Fe {
    u64 l[4];

    static Fe zero() { Fe r; r.l[0] = r.l[1] = r.l[2] = r.l[3] = 0; return r; }
    static Fe one() { Fe r; memcpy(r.l, P::R, 32); return r; }
    static Fe load(const u64* p) { Fe r; memcpy(r.l, p, 32); return r; }
    void store(u64* p) const { memcpy(p, l, 32); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const Fe& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }

    static bool geq_mod(const u64* a) {
        for (int i = 3; i >= 0; --i) {
            if (a[i] > P::M[i]) return true;
            if (a[i] < P::M[i]) return false;
        }
        return true;
    }
    static void sub_mod_inplace(u64* a) {
        u64 borrow = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)a[i] - P::M[i] - borrow;
            a[i] = (u64)d;
            borrow = (u64)(d >> 64) & 1;
        }
    }
    Fe add(const Fe& o) const {
        Fe r; u64 carry = 0;
        for (int i = 0; i < 4; ++i) {
            u128 s = (u128)l[i] + o.l[i] + carry;
            r.l[i] = (u64)s; carry = (u64)(s >> 64);
        }
        // moduli are < 2^255 so no carry out of 256 bits
        if (geq_mod(r.l)) sub_mod_inplace(r.l);
        return r;
    }
    Fe sub(const Fe& o) const {
        Fe r; u64 borrow = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)l[i] - o.l[i] - borrow;
            r.l[i] = (u64)d; borrow = (u64)(d >> 64) & 1;
        }
        if (borrow) {
            u64 carry = 0;
            for (int i = 0; i < 4; ++i) {
                u128 s = (u128)r.l[i] + P::M[i] + carry;
                r.l[i] = (u64)s; carry = (u64)(s >> 64);
            }
        }
        return r;
    }
    Fe neg() const { return is_zero() ? *this : zero().sub(*this); }
    Fe dbl() const { return add(*this); }

    // pasta: schoolbook 4x4 then montgomery_reduce; CIOS here (same function mod m).
    Fe mul(const Fe& o) const {
        u64 t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u64 carry = 0;
            for (int j = 0; j < 4; ++j) {
                u128 s = (u128)l[j] * o.l[i] + t[j] + carry;
                t[j] = (u64)s; carry = (u64)(s >> 64);
            }
            u128 s = (u128)t[4] + carry;
            t[4] = (u64)s; t[5] = (u64)(s >> 64);
            u64 m = t[0] * P::INV;
            s = (u128)m * P::M[0] + t[0];
            carry = (u64)(s >> 64);
            for (int j = 1; j < 4; ++j) {
                s = (u128)m * P::M[j] + t[j] + carry;
                t[j - 1] = (u64)s; carry = (u64)(s >> 64);
            }
            s = (u128)t[4] + carry;
            t[3] = (u64)s;
            t[4] = t[5] + (u64)(s >> 64);
        }
        Fe r; memcpy(r.l, t, 32);
        if (t[4] || geq_mod(r.l)) sub_mod_inplace(r.l);
        return r;
    }
    Fe sqr() const { return mul(*this); }

    Fe pow_vartime(const u64 e[4]) const {
        Fe r = one();
        for (int i = 255; i >= 0; --i) {
            r = r.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) r = r.mul(*this);
        }
        return r;
    }
    Fe inv() const {  // a^(m-2); inv(0) = 0
        u64 e[4] = {P::M[0] - 2, P::M[1], P::M[2], P::M[3]};
        return pow_vartime(e);
    }
    Fe to_mont() const { Fe r2 = load(P::R2); return mul(r2); }
    Fe from_mont() const { Fe o = zero(); o.l[0] = 1; return mul(o); }  // == to_repr() limbs
};

// ---- curve y^2 = x^3 + 5 over Fe<P>, Jacobian ----------------------------------------
template <class P>
struct Aff { Fe<P> x, y; bool inf; };

template <class P>
struct Jac {
    Fe<P> X, Y, Z;
    static Jac identity() { Jac r; r.X = Fe<P>::zero(); r.Y = Fe<P>::zero(); r.Z = Fe<P>::zero(); return r; }
    static Jac from_affine(const Aff<P>& a) {
        if (a.inf) return identity();
        Jac r; r.X = a.x; r.Y = a.y; r.Z = Fe<P>::one(); return r;
    }
    bool is_identity() const { return Z.is_zero(); }

    Jac dbl() const {  // dbl-2009-l (a = 0)
        if (is_identity()) return *this;
        Fe<P> A = X.sqr(), B = Y.sqr(), C = B.sqr();
        Fe<P> D = X.add(B).sqr().sub(A).sub(C).dbl();
        Fe<P> E = A.dbl().add(A), F = E.sqr();
        Jac r;
        r.Z = Y.mul(Z).dbl();
        r.X = F.sub(D.dbl());
        r.Y = E.mul(D.sub(r.X)).sub(C.dbl().dbl().dbl());
        return r;
    }
    Jac add(const Jac& o) const {  // add-2007-bl with the complete case analysis
        if (is_identity()) return o;
        if (o.is_identity()) return *this;
        Fe<P> Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
        Fe<P> U1 = X.mul(Z2Z2), U2 = o.X.mul(Z1Z1);
        Fe<P> S1 = Y.mul(Z2Z2).mul(o.Z), S2 = o.Y.mul(Z1Z1).mul(Z);
        if (U1 == U2) {
            if (S1 == S2) return dbl();
            return identity();
        }
        Fe<P> H = U2.sub(U1), I = H.dbl().sqr(), J = H.mul(I);
        Fe<P> rr = S2.sub(S1).dbl(), V = U1.mul(I);
        Jac r;
        r.X = rr.sqr().sub(J).sub(V.dbl());
        r.Y = rr.mul(V.sub(r.X)).sub(S1.mul(J).dbl());
        r.Z = Z.add(o.Z).sqr().sub(Z1Z1).sub(Z2Z2).mul(H);
        return r;
    }
    Jac add_mixed(const Aff<P>& o) const {  // madd-2007-bl
        if (o.inf) return *this;
        if (is_identity()) return from_affine(o);
        Fe<P> Z1Z1 = Z.sqr();
        Fe<P> U2 = o.x.mul(Z1Z1), S2 = o.y.mul(Z1Z1).mul(Z);
        if (X == U2) {
            if (Y == S2) return dbl();
            return identity();
        }
        Fe<P> H = U2.sub(X), HH = H.sqr(), I = HH.dbl().dbl(), J = H.mul(I);
        Fe<P> rr = S2.sub(Y).dbl(), V = X.mul(I);
        Jac r;
        r.X = rr.sqr().sub(J).sub(V.dbl());
        r.Y = rr.mul(V.sub(r.X)).sub(Y.mul(J).dbl());
        r.Z = Z.add(H).sqr().sub(Z1Z1).sub(HH);
        return r;
    }
    Aff<P> to_affine() const {
        Aff<P> a;
        if (is_identity()) { a.x = Fe<P>::zero(); a.y = Fe<P>::zero(); a.inf = true; return a; }
        Fe<P> zi = Z.inv(), zi2 = zi.sqr();
        a.x = X.mul(zi2); a.y = Y.mul(zi2).mul(zi); a.inf = false;
        return a;
    }
};

// 64-byte POD of the C ABI: x[4], y[4]; identity = all-zero (not on the curve)
template <class P>
Aff<P> load_aff(const u64* p) {
    Aff<P> a; a.x = Fe<P>::load(p); a.y = Fe<P>::load(p + 4);
    a.inf = a.x.is_zero() && a.y.is_zero();
    return a;
}
template <class P>
void store_aff(const Aff<P>& a, u64* p) {
    if (a.inf) { memset(p, 0, 64); return; }
    a.x.store(p); a.y.store(p + 4);
}
template <class P>
void store_jac(const Jac<P>& j, u64* p) { j.X.store(p); j.Y.store(p + 4); j.Z.store(p + 8); }
template <class P>
Jac<P> load_jac(const u64* p) { Jac<P> j; j.X = Fe<P>::load(p); j.Y = Fe<P>::load(p + 4); j.Z = Fe<P>::load(p + 8); return j; }

// ---- multiexp_serial / best_multiexp (arithmetic.rs) ----------------------------------
// PS = scalar-field params, PB = base-field params
template <class PS, class PB>
void multiexp_serial(const u64* coeffs, const u64* bases, size_t n, Jac<PB>& acc) {
    // coeffs.iter().map(|a| a.to_repr())
    std::vector<u64> reprs(n * 4);
    for (size_t i = 0; i < n; ++i) Fe<PS>::load(coeffs + 4 * i).from_mont().store(&reprs[4 * i]);
    const unsigned char* bytes = (const unsigned char*)reprs.data();

    size_t c;
    if (n < 4) c = 1;
    else if (n < 32) c = 3;
    else c = (size_t)std::ceil(std::log((double)n));
    const size_t segments = 256 / c + 1;

    auto get_at = [&](size_t segment, const unsigned char* b) -> size_t {
        size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
        if (skip_bytes >= 32) return 0;
        unsigned char v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        size_t len = 32 - skip_bytes; if (len > 8) len = 8;
        memcpy(v, b + skip_bytes, len);
        u64 tmp; memcpy(&tmp, v, 8);
        tmp >>= (skip_bits - skip_bytes * 8);
        return (size_t)(tmp % ((u64)1 << c));
    };

    // bucket enum: 0 None, 1 Affine, 2 Projective
    struct Bucket { unsigned char kind; Aff<PB> a; Jac<PB> p; };
    const size_t nb = ((size_t)1 << c) - 1;
    std::vector<Bucket> buckets(nb);

    for (size_t seg = segments; seg-- > 0;) {
        for (size_t k = 0; k < c; ++k) acc = acc.dbl();
        for (size_t b = 0; b < nb; ++b) buckets[b].kind = 0;
        for (size_t i = 0; i < n; ++i) {
            size_t d = get_at(seg, bytes + 32 * i);
            if (d == 0) continue;
            Bucket& bk = buckets[d - 1];
            Aff<PB> base = load_aff<PB>(bases + 8 * i);
            if (bk.kind == 0) { bk.kind = 1; bk.a = base; }
            else if (bk.kind == 1) { bk.p = Jac<PB>::from_affine(bk.a).add_mixed(base); bk.kind = 2; }
            else bk.p = bk.p.add_mixed(base);
        }
        Jac<PB> running = Jac<PB>::identity();
        for (size_t b = nb; b-- > 0;) {
            Bucket& bk = buckets[b];
            if (bk.kind == 1) running = running.add_mixed(bk.a);
            else if (bk.kind == 2) running = running.add(bk.p);
            acc = acc.add(running);
        }
    }
}

template <class PS, class PB>
void best_multiexp(const u64* coeffs, const u64* bases, size_t n, int threads, u64* out_xyz) {
    if (threads < 1) threads = 1;
    Jac<PB> total = Jac<PB>::identity();
    if (n > (size_t)threads && threads > 1) {
        size_t chunk = n / threads;
        size_t nchunks = (n + chunk - 1) / chunk;
        std::vector<Jac<PB>> res(nchunks, Jac<PB>::identity());
        std::vector<std::thread> th;
        for (size_t k = 0; k < nchunks; ++k) {
            size_t lo = k * chunk, len = (lo + chunk <= n) ? chunk : n - lo;
            th.emplace_back([=, &res] { multiexp_serial<PS, PB>(coeffs + 4 * lo, bases + 8 * lo, len, res[k]); });
        }
        for (auto& t : th) t.join();
        for (auto& r : res) total = total.add(r);
    } else {
        multiexp_serial<PS, PB>(coeffs, bases, n, total);
    }
    store_jac(total, out_xyz);
}

// ---- best_fft (arithmetic.rs) -----------------------------------------------------------
static inline uint32_t bitreverse32(uint32_t n, uint32_t l) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < l; ++i) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}

template <class P>
void recursive_butterfly(Fe<P>* a, size_t n, size_t twiddle_chunk, const Fe<P>* tw, int depth_par) {
    if (n == 2) {
        Fe<P> t = a[1];
        a[1] = a[0].sub(t);
        a[0] = a[0].add(t);
        return;
    }
    size_t half = n / 2;
    if (depth_par > 0) {
        std::thread th([=] { recursive_butterfly<P>(a, half, twiddle_chunk * 2, tw, depth_par - 1); });
        recursive_butterfly<P>(a + half, half, twiddle_chunk * 2, tw, depth_par - 1);
        th.join();
    } else {
        recursive_butterfly<P>(a, half, twiddle_chunk * 2, tw, 0);
        recursive_butterfly<P>(a + half, half, twiddle_chunk * 2, tw, 0);
    }
    // case k = 0: twiddle is one
    Fe<P> t = a[half];
    a[half] = a[0].sub(t);
    a[0] = a[0].add(t);
    for (size_t k = 1; k < half; ++k) {
        Fe<P> t2 = a[half + k].mul(tw[k * twiddle_chunk]);
        a[half + k] = a[k].sub(t2);
        a[k] = a[k].add(t2);
    }
}

template <class P>
void best_fft(u64* data, const u64* omega_limbs, uint32_t log_n, int threads) {
    Fe<P>* a = (Fe<P>*)data;
    size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n; ++k) {
        size_t rk = bitreverse32((uint32_t)k, log_n);
        if (k < rk) { Fe<P> t = a[k]; a[k] = a[rk]; a[rk] = t; }
    }
    if (log_n == 0) return;
    Fe<P> omega = Fe<P>::load(omega_limbs);
    std::vector<Fe<P>> tw(n / 2 ? n / 2 : 1);
    Fe<P> w = Fe<P>::one();
    for (size_t i = 0; i < n / 2; ++i) { tw[i] = w; w = w.mul(omega); }
    int log_threads = 0;
    while ((1 << (log_threads + 1)) <= threads) ++log_threads;
    if (n == 1) return;
    recursive_butterfly<P>(a, n, 1, tw.data(), log_n > 10 ? log_threads : 0);
}

// ---- scalar mul (double-and-add over canonical scalar limbs) ----------------------------
template <class PB>
Jac<PB> scalar_mul(const Aff<PB>& base, const u64 k[4]) {
    Jac<PB> acc = Jac<PB>::identity();
    for (int i = 255; i >= 0; --i) {
        acc = acc.dbl();
        if ((k[i / 64] >> (i % 64)) & 1) acc = acc.add_mixed(base);
    }
    return acc;
}

template <class PB>
Aff<PB> generator() {
    Aff<PB> g; g.inf = false;
    Fe<PB> one = Fe<PB>::one();
    g.x = one.neg();
    g.y = one.dbl();
    return g;
}

// P_i = (s0 + i*d) * G for i in [lo, hi): start by scalar-mul, then repeated addition of D
template <class PB>
void gen_bases_range(u64 s0, u64 d, size_t lo, size_t hi, u64* out) {
    Aff<PB> G = generator<PB>();
    u128 k0 = (u128)s0 + (u128)lo * d;
    u64 k[4] = {(u64)k0, (u64)(k0 >> 64), 0, 0};
    u64 dk[4] = {d, 0, 0, 0};
    Jac<PB> cur = scalar_mul<PB>(G, k);
    Aff<PB> D = scalar_mul<PB>(G, dk).to_affine();
    // batch-normalise in blocks of 1024 (Montgomery's trick)
    const size_t B = 1024;
    std::vector<Jac<PB>> blk(B);
    std::vector<Fe<PB>> pref(B);
    for (size_t base = lo; base < hi; base += B) {
        size_t m = (hi - base < B) ? hi - base : B;
        for (size_t j = 0; j < m; ++j) { blk[j] = cur; cur = cur.add_mixed(D); }
        Fe<PB> acc = Fe<PB>::one();
        for (size_t j = 0; j < m; ++j) { pref[j] = acc; if (!blk[j].is_identity()) acc = acc.mul(blk[j].Z); }
        Fe<PB> inv = acc.inv();
        for (size_t j = m; j-- > 0;) {
            Aff<PB> a;
            if (blk[j].is_identity()) { a.inf = true; a.x = a.y = Fe<PB>::zero(); }
            else {
                Fe<PB> zi = inv.mul(pref[j]);
                inv = inv.mul(blk[j].Z);
                Fe<PB> zi2 = zi.sqr();
                a.x = blk[j].X.mul(zi2); a.y = blk[j].Y.mul(zi2).mul(zi); a.inf = false;
            }
            store_aff(a, out + 8 * (base - lo + j));
        }
    }
}

}  // namespace

// ======================================================================================
// C entry points (ctypes).  field: 0 = Fp, 1 = Fq.  curve: 0 = pallas (base Fp, scalar Fq),
// 1 = vesta (base Fq, scalar Fp).
// ======================================================================================
extern "C" {

enum { ORC_ADD = 0, ORC_SUB = 1, ORC_MUL = 2, ORC_SQR = 3, ORC_NEG = 4, ORC_INV = 5, ORC_TO_MONT = 6, ORC_FROM_MONT = 7 };

}  // extern "C"
template <class P>
static void field_op_t(int op, const u64* a, const u64* b, u64* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        Fe<P> x = Fe<P>::load(a + 4 * i), y = b ? Fe<P>::load(b + 4 * i) : Fe<P>::zero(), r;
        switch (op) {
            case ORC_ADD: r = x.add(y); break;
            case ORC_SUB: r = x.sub(y); break;
            case ORC_MUL: r = x.mul(y); break;
            case ORC_SQR: r = x.sqr(); break;
            case ORC_NEG: r = x.neg(); break;
            case ORC_INV: r = x.inv(); break;
            case ORC_TO_MONT: r = x.to_mont(); break;
            default: r = x.from_mont(); break;
        }
        r.store(out + 4 * i);
    }
}
extern "C" {
int orc_field_op(int field, int op, const u64* a, const u64* b, u64* out, size_t n) {
    if (field == 0) field_op_t<FpP>(op, a, b, out, n); else field_op_t<FqP>(op, a, b, out, n);
    return 0;
}

}  // extern "C"
template <class PB>
static void point_op_t(int op, const u64* p, const u64* q, u64* out) {
    Jac<PB> a = load_jac<PB>(p), r;
    if (op == 0) r = a.add(load_jac<PB>(q));
    else if (op == 1) r = a.add_mixed(load_aff<PB>(q));
    else r = a.dbl();
    store_jac(r, out);
}
extern "C" {
// op: 0 add (q Jacobian 12 limbs), 1 mixed add (q affine 8 limbs), 2 double
int orc_point_op(int curve, int op, const u64* p, const u64* q, u64* out) {
    if (curve == 0) point_op_t<FpP>(op, p, q, out); else point_op_t<FqP>(op, p, q, out);
    return 0;
}
int orc_to_affine(int curve, const u64* xyz, u64* xy) {
    if (curve == 0) store_aff(load_jac<FpP>(xyz).to_affine(), xy);
    else store_aff(load_jac<FqP>(xyz).to_affine(), xy);
    return 0;
}
// scalar: canonical 4 limbs
int orc_scalar_mul(int curve, const u64* base_xy, const u64* k, u64* out_xyz) {
    if (curve == 0) store_jac(scalar_mul<FpP>(load_aff<FpP>(base_xy), k), out_xyz);
    else store_jac(scalar_mul<FqP>(load_aff<FqP>(base_xy), k), out_xyz);
    return 0;
}
// coeffs: n x 4 Montgomery limbs of the curve's scalar field; bases: n x 8; out: Jacobian 12 limbs
int orc_best_multiexp(int curve, const u64* coeffs, const u64* bases, size_t n, int threads, u64* out_xyz) {
    if (curve == 0) best_multiexp<FqP, FpP>(coeffs, bases, n, threads, out_xyz);
    else best_multiexp<FpP, FqP>(coeffs, bases, n, threads, out_xyz);
    return 0;
}
int orc_best_fft(int field, u64* a, const u64* omega, uint32_t log_n, int threads) {
    if (field == 0) best_fft<FpP>(a, omega, log_n, threads); else best_fft<FqP>(a, omega, log_n, threads);
    return 0;
}
// out: n x 8 limbs, P_i = (s0 + i*d) * G
int orc_gen_bases(int curve, u64 s0, u64 d, size_t n, int threads, u64* out) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> th;
    size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per > n ? n : lo + per;
        if (lo >= hi) break;
        th.emplace_back([=] {
            if (curve == 0) gen_bases_range<FpP>(s0, d, lo, hi, out + 8 * lo);
            else gen_bases_range<FqP>(s0, d, lo, hi, out + 8 * lo);
        });
    }
    for (auto& t : th) t.join();
    return 0;
}
int orc_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"

// ======================================================================================
// Round-2 additions: the callers either side of best_multiexp / best_fft at REAL sizes
// (k = 10 .. 18), so that the GPU path is compared with a restatement there and not only
// at k <= 6 (Python big-int).  Same status as the rest of this file: test infrastructure,
// parity unpinned by the reference, cross-checked against oracle/pasta.py at small k in
// tests/test_oracle.py.
// ======================================================================================
namespace {

template <class F>
void parallel_for(size_t n, int threads, F&& body) {  // body(lo, hi)
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    if (threads == 1) { body((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per > n ? n : lo + per;
        if (lo >= hi) break;
        th.emplace_back([=, &body] { body(lo, hi); });
    }
    for (auto& t : th) t.join();
}

// C::Curve::batch_normalize: Jacobian -> affine with one inversion per block (Montgomery's trick)
template <class PB>
void batch_normalize(const Jac<PB>* in, size_t n, u64* out_xy, int threads) {
    parallel_for(n, threads, [&](size_t lo, size_t hi) {
        const size_t B = 1024;
        std::vector<Fe<PB>> pref(B);
        for (size_t base = lo; base < hi; base += B) {
            const size_t m = hi - base < B ? hi - base : B;
            Fe<PB> acc = Fe<PB>::one();
            for (size_t j = 0; j < m; ++j) { pref[j] = acc; if (!in[base + j].is_identity()) acc = acc.mul(in[base + j].Z); }
            Fe<PB> inv = acc.inv();
            for (size_t j = m; j-- > 0;) {
                const Jac<PB>& p = in[base + j];
                Aff<PB> a;
                if (p.is_identity()) { a.inf = true; a.x = a.y = Fe<PB>::zero(); }
                else {
                    Fe<PB> zi = inv.mul(pref[j]);
                    inv = inv.mul(p.Z);
                    Fe<PB> zi2 = zi.sqr();
                    a.x = p.X.mul(zi2); a.y = p.Y.mul(zi2).mul(zi); a.inf = false;
                }
                store_aff(a, out_xy + 8 * (base + j));
            }
        }
    });
}

// `point * scalar` (pasta: double-and-add over the canonical bits of the scalar), scalar given in Montgomery form
template <class PS, class PB>
Jac<PB> scale_point(const Jac<PB>& p, const Fe<PS>& s_mont) {
    u64 k[4];
    s_mont.from_mont().store(k);
    Jac<PB> acc = Jac<PB>::identity();
    for (int i = 255; i >= 0; --i) {
        acc = acc.dbl();
        if ((k[i / 64] >> (i % 64)) & 1) acc = acc.add(p);
    }
    return acc;
}

// ---- unstructured bases: P_i = h_i * G with h_i = four SplitMix64 words of (seed, i) reduced below 2^254 ---------------
// (fixed-base comb over a byte table of G: 32 mixed additions per point instead of a 255-bit double-and-add; the group
// element is the definition's)
static inline u64 splitmix(u64& s) { u64 z = (s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
static inline void hashed_scalar(u64 seed, size_t i, u64 k[4]) {
    u64 s = seed ^ (0xD1B54A32D192ED03ULL * (u64)(i + 1));
    k[0] = splitmix(s); k[1] = splitmix(s); k[2] = splitmix(s); k[3] = splitmix(s) >> 2;
}
template <class PB>
void gen_bases_hashed(u64 seed, size_t n, int threads, u64* out) {
    // table[j][d] = d * 2^(8 j) * G, d in 1..255, affine
    std::vector<Jac<PB>> tj(32 * 255);
    Jac<PB> base = Jac<PB>::from_affine(generator<PB>());
    for (int j = 0; j < 32; ++j) {
        Jac<PB> cur = base;
        for (int d = 1; d <= 255; ++d) { tj[j * 255 + d - 1] = cur; cur = cur.add(base); }
        base = cur;  // 256 * previous base
    }
    std::vector<u64> txy(8 * tj.size());
    batch_normalize<PB>(tj.data(), tj.size(), txy.data(), threads);
    std::vector<Jac<PB>> pts(n);
    parallel_for(n, threads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            u64 k[4];
            hashed_scalar(seed, i, k);
            Jac<PB> acc = Jac<PB>::identity();
            for (int j = 0; j < 32; ++j) {
                const unsigned d = (unsigned)(k[j / 8] >> (8 * (j % 8))) & 0xffu;
                if (d) acc = acc.add_mixed(load_aff<PB>(&txy[8 * (j * 255 + d - 1)]));
            }
            pts[i] = acc;
        }
    });
    batch_normalize<PB>(pts.data(), n, out, threads);
}

// ---- best_fft::<C::Curve> (arithmetic.rs): the same butterflies as for field elements, group_scale = point * scalar ----
// points: n affine PODs in place; omega: scalar field, Montgomery.  Output normalised to affine (what Params::new does next).
template <class PS, class PB>
void best_fft_points(u64* xy, const u64* omega_limbs, uint32_t log_n, int threads) {
    const size_t n = (size_t)1 << log_n;
    std::vector<Jac<PB>> a(n);
    for (size_t i = 0; i < n; ++i) a[i] = Jac<PB>::from_affine(load_aff<PB>(xy + 8 * i));
    for (size_t k = 0; k < n; ++k) {
        size_t rk = bitreverse32((uint32_t)k, log_n);
        if (k < rk) std::swap(a[k], a[rk]);
    }
    const Fe<PS> omega = Fe<PS>::load(omega_limbs);
    std::vector<Fe<PS>> tw(n / 2 ? n / 2 : 1);
    Fe<PS> w = Fe<PS>::one();
    for (size_t i = 0; i < n / 2; ++i) { tw[i] = w; w = w.mul(omega); }
    size_t chunk = 2, twiddle_chunk = n / 2;
    for (uint32_t st = 0; st < log_n; ++st) {
        const size_t half = chunk / 2, nb = n / 2;  // nb butterflies per stage, all independent
        parallel_for(nb, threads, [&](size_t lo, size_t hi) {
            for (size_t q = lo; q < hi; ++q) {
                const size_t blk = q / half, i = q % half;
                Jac<PB>& x = a[blk * chunk + i];
                Jac<PB>& y = a[blk * chunk + half + i];
                Jac<PB> t = i == 0 ? y : scale_point<PS, PB>(y, tw[i * twiddle_chunk]);  // "case when twiddle factor is one"
                Jac<PB> neg = t; neg.Y = neg.Y.neg();
                y = x.add(neg);
                x = x.add(t);
            }
        });
        chunk *= 2; twiddle_chunk /= 2;
    }
    batch_normalize<PB>(a.data(), n, xy, threads);
}

// ---- poly::commitment::prover::create_proof (the IPA opening), restated literally: G' is materialised and collapsed
//      with n / 2 + n / 4 + ... scalar multiplications per proof, the round MSMs run through best_multiexp above ----------
struct OrcTranscript {
    void* ctx;
    void (*write_point)(void*, const u64* xyz);   // normalised Jacobian, Z = 1 (identity: all zero)
    void (*write_scalar)(void*, const u64* s);    // Montgomery limbs
    void (*squeeze)(void*, u64* out);             // challenge scalar, Montgomery limbs
};
typedef void (*OrcRng)(void*, u64* out);

template <class PB>
void write_point_norm(const OrcTranscript* tr, const Jac<PB>& p) {
    u64 xyz[12];
    const Aff<PB> a = p.to_affine();
    if (a.inf) memset(xyz, 0, sizeof(xyz));
    else { a.x.store(xyz); a.y.store(xyz + 4); Fe<PB>::one().store(xyz + 8); }
    tr->write_point(tr->ctx, xyz);
}

template <class PS, class PB>
Jac<PB> multiexp_j(const Fe<PS>* coeffs, const u64* bases, size_t n, int threads) {
    u64 out[12];
    best_multiexp<PS, PB>((const u64*)coeffs, bases, n, threads, out);
    return load_jac<PB>(out);
}

template <class PS>
Fe<PS> eval_poly(const std::vector<Fe<PS>>& p, const Fe<PS>& x) {
    Fe<PS> acc = Fe<PS>::zero();
    for (size_t i = p.size(); i-- > 0;) acc = acc.mul(x).add(p[i]);
    return acc;
}
template <class PS>
Fe<PS> inner_product(const Fe<PS>* a, const Fe<PS>* b, size_t n) {
    Fe<PS> acc = Fe<PS>::zero();
    for (size_t i = 0; i < n; ++i) acc = acc.add(a[i].mul(b[i]));
    return acc;
}

template <class PS, class PB>
int ipa_create_proof(uint32_t k, const u64* g_xy, const u64* w_xy, const u64* u_xy, const u64* p_poly, const u64* p_blind_l, const u64* x3_l,
                     const u64* s_poly_l, const u64* s_blind_l, const OrcTranscript* tr, OrcRng rng, void* rng_ctx, int threads, u64* out_c, u64* out_f) {
    const size_t n = (size_t)1 << k;
    const Fe<PS> x3 = Fe<PS>::load(x3_l), p_blind = Fe<PS>::load(p_blind_l), s_blind = Fe<PS>::load(s_blind_l);
    std::vector<Fe<PS>> s_poly(n), p_prime(n), b(n);
    for (size_t i = 0; i < n; ++i) s_poly[i] = Fe<PS>::load(s_poly_l + 4 * i);
    s_poly[0] = s_poly[0].sub(eval_poly(s_poly, x3));  // s(x3) = 0
    {   // params.commit(&s_poly, s_poly_blind): multiexp over g || w
        std::vector<u64> bases(8 * (n + 1));
        memcpy(bases.data(), g_xy, 64 * n);
        memcpy(bases.data() + 8 * n, w_xy, 64);
        std::vector<Fe<PS>> sc(s_poly);
        sc.push_back(s_blind);
        write_point_norm(tr, multiexp_j<PS, PB>(sc.data(), bases.data(), n + 1, threads));
    }
    u64 tmp[4];
    tr->squeeze(tr->ctx, tmp); const Fe<PS> xi = Fe<PS>::load(tmp);
    tr->squeeze(tr->ctx, tmp); const Fe<PS> z = Fe<PS>::load(tmp);
    for (size_t i = 0; i < n; ++i) p_prime[i] = s_poly[i].mul(xi).add(Fe<PS>::load(p_poly + 4 * i));
    const Fe<PS> v = eval_poly(p_prime, x3);
    p_prime[0] = p_prime[0].sub(v);
    Fe<PS> f = s_blind.mul(xi).add(p_blind);
    { Fe<PS> cur = Fe<PS>::one(); for (size_t i = 0; i < n; ++i) { b[i] = cur; cur = cur.mul(x3); } }
    std::vector<u64> g_prime(g_xy, g_xy + 8 * n);
    u64 uw[16];
    memcpy(uw, u_xy, 64); memcpy(uw + 8, w_xy, 64);
    for (uint32_t j = 0; j < k; ++j) {
        const size_t half = (size_t)1 << (k - j - 1);
        Jac<PB> l_j = multiexp_j<PS, PB>(&p_prime[half], g_prime.data(), half, threads);
        Jac<PB> r_j = multiexp_j<PS, PB>(&p_prime[0], g_prime.data() + 8 * half, half, threads);
        const Fe<PS> value_l = inner_product(&p_prime[half], &b[0], half), value_r = inner_product(&p_prime[0], &b[half], half);
        rng(rng_ctx, tmp); const Fe<PS> l_rand = Fe<PS>::load(tmp);
        rng(rng_ctx, tmp); const Fe<PS> r_rand = Fe<PS>::load(tmp);
        { Fe<PS> sc[2] = {value_l.mul(z), l_rand}; l_j = l_j.add(multiexp_j<PS, PB>(sc, uw, 2, 1)); }
        { Fe<PS> sc[2] = {value_r.mul(z), r_rand}; r_j = r_j.add(multiexp_j<PS, PB>(sc, uw, 2, 1)); }
        write_point_norm(tr, l_j);
        write_point_norm(tr, r_j);
        tr->squeeze(tr->ctx, tmp);
        const Fe<PS> u_j = Fe<PS>::load(tmp);
        if (u_j.is_zero()) return -1;  // u_j.invert().unwrap()
        const Fe<PS> u_inv = u_j.inv();
        for (size_t i = 0; i < half; ++i) {
            p_prime[i] = p_prime[i].add(p_prime[i + half].mul(u_inv));
            b[i] = b[i].add(b[i + half].mul(u_j));
        }
        p_prime.resize(half); b.resize(half);
        {   // parallel_generator_collapse(&mut g_prime, u_j); truncate; batch_normalize
            std::vector<Jac<PB>> col(half);
            parallel_for(half, threads, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    const Jac<PB> hi_pt = Jac<PB>::from_affine(load_aff<PB>(&g_prime[8 * (i + half)]));
                    col[i] = scale_point<PS, PB>(hi_pt, u_j).add_mixed(load_aff<PB>(&g_prime[8 * i]));
                }
            });
            g_prime.resize(8 * half);
            batch_normalize<PB>(col.data(), half, g_prime.data(), threads);
        }
        f = f.add(l_rand.mul(u_inv)).add(r_rand.mul(u_j));
    }
    u64 c_l[4], f_l[4];
    p_prime[0].store(c_l); f.store(f_l);
    tr->write_scalar(tr->ctx, c_l);
    tr->write_scalar(tr->ctx, f_l);
    if (out_c) memcpy(out_c, c_l, 32);
    if (out_f) memcpy(out_f, f_l, 32);
    return 0;
}

}  // namespace

extern "C" {

// out: n x 8 limbs, P_i = h_i * G with h_i a hash of (seed, i): no arithmetic structure between the bases
int orc_gen_bases_hashed(int curve, u64 seed, size_t n, int threads, u64* out) {
    if (curve == 0) gen_bases_hashed<FpP>(seed, n, threads, out); else gen_bases_hashed<FqP>(seed, n, threads, out);
    return 0;
}
// the discrete logs of the above (canonical limbs), for closed-form checks
int orc_hashed_scalars(u64 seed, size_t n, u64* out_canonical) {
    for (size_t i = 0; i < n; ++i) hashed_scalar(seed, i, out_canonical + 4 * i);
    return 0;
}
// best_fft over curve points in place (affine in, affine out); omega: the curve's scalar field, Montgomery
int orc_best_fft_points(int curve, u64* xy, const u64* omega, uint32_t log_n, int threads) {
    if (curve == 0) best_fft_points<FqP, FpP>(xy, omega, log_n, threads); else best_fft_points<FpP, FqP>(xy, omega, log_n, threads);
    return 0;
}
// out[i] = scalars[i] * base (scalars Montgomery), affine
int orc_scale_points(int curve, const u64* base_xy, const u64* scalars_mont, size_t n, int threads, u64* out_xy) {
    if (curve == 0) {
        std::vector<Jac<FpP>> r(n);
        const Jac<FpP> b = Jac<FpP>::from_affine(load_aff<FpP>(base_xy));
        parallel_for(n, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) r[i] = scale_point<FqP, FpP>(b, Fe<FqP>::load(scalars_mont + 4 * i)); });
        batch_normalize<FpP>(r.data(), n, out_xy, threads);
    } else {
        std::vector<Jac<FqP>> r(n);
        const Jac<FqP> b = Jac<FqP>::from_affine(load_aff<FqP>(base_xy));
        parallel_for(n, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) r[i] = scale_point<FpP, FqP>(b, Fe<FpP>::load(scalars_mont + 4 * i)); });
        batch_normalize<FqP>(r.data(), n, out_xy, threads);
    }
    return 0;
}
// pointwise base[i] * scalars[i] (the generator-collapse primitive), affine out
int orc_scale_points_each(int curve, const u64* bases_xy, const u64* scalars_mont, size_t n, int threads, u64* out_xy) {
    if (curve == 0) {
        std::vector<Jac<FpP>> r(n);
        parallel_for(n, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) r[i] = scale_point<FqP, FpP>(Jac<FpP>::from_affine(load_aff<FpP>(bases_xy + 8 * i)), Fe<FqP>::load(scalars_mont + 4 * i)); });
        batch_normalize<FpP>(r.data(), n, out_xy, threads);
    } else {
        std::vector<Jac<FqP>> r(n);
        parallel_for(n, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) r[i] = scale_point<FpP, FqP>(Jac<FqP>::from_affine(load_aff<FqP>(bases_xy + 8 * i)), Fe<FpP>::load(scalars_mont + 4 * i)); });
        batch_normalize<FqP>(r.data(), n, out_xy, threads);
    }
    return 0;
}
int orc_ipa_create_proof(int curve, uint32_t k, const u64* g_xy, const u64* w_xy, const u64* u_xy, const u64* p_poly, const u64* p_blind, const u64* x3,
                         const u64* s_poly, const u64* s_blind, void* tr_ctx, void (*write_point)(void*, const u64*), void (*write_scalar)(void*, const u64*),
                         void (*squeeze)(void*, u64*), void (*rng)(void*, u64*), void* rng_ctx, int threads, u64* out_c, u64* out_f) {
    OrcTranscript tr{tr_ctx, write_point, write_scalar, squeeze};
    if (curve == 0) return ipa_create_proof<FqP, FpP>(k, g_xy, w_xy, u_xy, p_poly, p_blind, x3, s_poly, s_blind, &tr, rng, rng_ctx, threads, out_c, out_f);
    return ipa_create_proof<FpP, FqP>(k, g_xy, w_xy, u_xy, p_poly, p_blind, x3, s_poly, s_blind, &tr, rng, rng_ctx, threads, out_c, out_f);
}
// arithmetic::eval_polynomial / kate_division / the x-power folds of multiopen, over Montgomery limb arrays
int orc_eval_polynomial(int field, const u64* poly, size_t n, const u64* x, u64* out) {
    if (field == 0) { std::vector<Fe<FpP>> p(n); for (size_t i = 0; i < n; ++i) p[i] = Fe<FpP>::load(poly + 4 * i); eval_poly(p, Fe<FpP>::load(x)).store(out); }
    else { std::vector<Fe<FqP>> p(n); for (size_t i = 0; i < n; ++i) p[i] = Fe<FqP>::load(poly + 4 * i); eval_poly(p, Fe<FqP>::load(x)).store(out); }
    return 0;
}

}  // extern "C"
