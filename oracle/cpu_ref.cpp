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
