// CPU unit test of the host-side pieces of include/trh.hpp that need no device: the field arithmetic used for expression
// constants and checks, and the lowering of Expression trees to the stack program (stack discipline, Sethi-Ullman order,
// SUB / SQR selection).  Cross-checked against csrc/hostcombine.h (an independent 4 x 64-bit implementation) and against an
// interpreter of the program written here.  Built and run by tests/test_hostcombine.py (g++ only, never calls libtrh).
#include <cstdio>
#include <vector>
#include "../../include/trh.hpp"
#include "../../tiny-ram-halo2_amd/csrc/field.h"
#include "../../tiny-ram-halo2_amd/csrc/hostcombine.h"

using namespace trh;

static uint64_t seed = 0x243f6a8885a308d3ull;
static uint64_t next() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; }
static Limbs element() { return Limbs{next(), next(), next(), next() >> 2}; }

template <class F>
static int field_checks(Field f, const char* name) {
    int bad = 0;
    namespace hc = trh::hostcombine;
    for (int t = 0; t < 2000; ++t) {
        const Limbs a = element(), b = element();
        hc::H ha, hb;
        memcpy(&ha, a.data(), 32); memcpy(&hb, b.data(), 32);
        const hc::H hm = hc::mul<F>(ha, hb), hs = hc::add<F>(ha, hb), hd = hc::sub<F>(ha, hb);
        const Limbs m = host::mul(f, a, b), s = host::add(f, a, b), d = host::sub(f, a, b);
        if (memcmp(&hm, m.data(), 32) || memcmp(&hs, s.data(), 32) || memcmp(&hd, d.data(), 32)) ++bad;
    }
    const Limbs one = host::one(f);
    if (host::mul(f, one, one) != one) ++bad;
    if (host::mul(f, host::from_u64(f, 6), host::from_u64(f, 7)) != host::from_u64(f, 42)) ++bad;
    if (host::add(f, host::neg(f, host::from_u64(f, 5)), host::from_u64(f, 5)) != Limbs{0, 0, 0, 0}) ++bad;
    if (bad) std::printf("%s: %d field mismatches\n", name, bad);
    return bad;
}

// reference semantics of the stack program (what csrc/expr.hip executes per row)
static Limbs run_program(Field f, const Program& p, const std::vector<Limbs>& column_values /* one value per program column, rotation ignored */) {
    std::vector<Limbs> st;
    Limbs acc{0, 0, 0, 0};
    for (const trh_expr_insn_t& in : p.insns) {
        switch (in.op) {
            case TRH_EXPR_PUSH_COLUMN: st.push_back(column_values[in.a]); break;
            case TRH_EXPR_PUSH_CONST: st.push_back(p.consts[in.a]); break;
            case TRH_EXPR_ADD: { Limbs t = st.back(); st.pop_back(); st.back() = host::add(f, st.back(), t); break; }
            case TRH_EXPR_SUB: { Limbs t = st.back(); st.pop_back(); st.back() = host::sub(f, st.back(), t); break; }
            case TRH_EXPR_MUL: { Limbs t = st.back(); st.pop_back(); st.back() = host::mul(f, st.back(), t); break; }
            case TRH_EXPR_NEG: st.back() = host::neg(f, st.back()); break;
            case TRH_EXPR_SQR: st.back() = host::mul(f, st.back(), st.back()); break;
            case TRH_EXPR_MUL_CONST: st.back() = host::mul(f, st.back(), p.consts[in.a]); break;
            case TRH_EXPR_FOLD: acc = host::add(f, host::mul(f, acc, p.consts[in.a]), st.back()); st.pop_back(); break;
            case TRH_EXPR_STORE_ACC: return acc;
            default: std::printf("unexpected opcode %u\n", in.op); return Limbs{~0ull, 0, 0, 0};
        }
    }
    return acc;
}
static Limbs eval_tree(Field f, const Expr& e, const Program& p, const std::vector<Limbs>& vals) {
    switch (e->kind) {
        case Expression::Constant: return e->value;
        case Expression::Negated: return host::neg(f, eval_tree(f, e->a, p, vals));
        case Expression::Scaled: return host::mul(f, eval_tree(f, e->a, p, vals), e->value);
        case Expression::Sum: return host::add(f, eval_tree(f, e->a, p, vals), eval_tree(f, e->b, p, vals));
        case Expression::Product: return host::mul(f, eval_tree(f, e->a, p, vals), eval_tree(f, e->b, p, vals));
        default:
            for (size_t i = 0; i < p.columns.size(); ++i) if (p.columns[i].first == e->kind && p.columns[i].second == e->column) return vals[i];
            return Limbs{0, 0, 0, 0};
    }
}
static Expr random_tree(Field f, int depth) {
    if (depth == 0 || next() % 7 == 0) {
        const uint64_t k = next() % 4;
        if (k == 0) return constant(element());
        return k == 1 ? advice((uint32_t)(next() % 3)) : k == 2 ? fixed((uint32_t)(next() % 2)) : selector(0);
    }
    const uint64_t k = next() % 6;
    if (k == 0) return -random_tree(f, depth - 1);
    if (k == 1) return scaled(random_tree(f, depth - 1), element());
    Expr a = random_tree(f, depth - 1), b = random_tree(f, (int)(next() % (uint64_t)depth));
    if (next() & 1) std::swap(a, b);
    if (k == 2) return a - b;
    if (k == 3) return a * a;  // squaring of a shared node
    return k == 4 ? a + b : a * b;
}
static int lowering_checks(Field f, const char* name) {
    int bad = 0;
    for (int t = 0; t < 300; ++t) {
        std::vector<Expr> gates;
        for (uint64_t g = 0, ng = 1 + next() % 4; g < ng; ++g) gates.push_back(random_tree(f, 1 + (int)(next() % 6)));
        const Limbs y = element();
        const Program p = compile_gates(f, gates, y);
        std::vector<Limbs> vals(p.columns.size());
        for (auto& v : vals) v = element();
        Limbs want{0, 0, 0, 0};
        for (const Expr& g : gates) want = host::add(f, host::mul(f, want, y), eval_tree(f, g, p, vals));
        if (run_program(f, p, vals) != want) ++bad;
    }
    if (bad) std::printf("%s: %d lowering mismatches\n", name, bad);
    return bad;
}

int main() {
    int bad = field_checks<FpParams>(Field::Fp, "fp") + field_checks<FqParams>(Field::Fq, "fq");
    bad += lowering_checks(Field::Fp, "fp") + lowering_checks(Field::Fq, "fq");
    std::printf("trh.hpp host side: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
