// Pointwise evaluation of gate expressions over resident columns: the h(X) numerator of
// halo2_proofs 0.2.0 `plonk::create_proof` (plonk/prover.rs builds, for every gate polynomial, an AST over the
// extended-domain cosets of the advice / fixed / instance columns and folds the values with the challenge y:
// h = h * y + gate; reached from /root/reference/src/test_utils.rs:41-49; the gates themselves are the
// `Expression`s of /root/reference/src/circuits/tables/exe.rs:147-498, logic.rs:125-185, sprod.rs:65-92;
// SURVEY.md section 8 row f-4).  After coeff_to_extended the ~500 extended columns (33 GB at k = 18) live in
// HBM; evaluating the gates where the columns are avoids shipping them back to the host.
//
// A compiled expression is a straight-line program for a small stack machine, one thread per row:
//   T, N      top / next of the evaluation stack, in registers
//   M[...]    deeper stack entries and user locals, in LDS (nine limbs per thread in three conflict-free planes);
//             which slot an instruction spills to / refills from is fixed when the program is created, because
//             every thread runs the same program (no stack pointer at run time)
//   ACC       accumulator of FOLD (acc = acc * const + T), the y-Horner over gates
// Column reads are `column[(row + rotation * rot_step) mod 2^log_n]` (Rotation(r) on the extended coset moves by
// r * 2^(extended_k - k) rows).  Integer work, HBM-bound for cheap gates: 32 B per column query.
//
// Arithmetic: the machine computes in the signed 29-bit lazy domain of field.h (Fy, Montgomery constant 2^261) -- a multiplication
// there is 126 multiply-adds + 54 instructions of carry handling against ~330 instructions for the canonical fe_mul (16-bit round,
// realignment, conditional subtraction), additions are one carry chain without a conditional subtraction.  A column word x 2^256
// becomes the domain value (x 2^261) by reading its limbs five bits lower (x 32, value < 32 m: free); constants are multiplied by
// 32 mod m on the host when the program is created; results leave through fy_to_fe (canonical Montgomery words, bit-exact).
// Values are only bounded, not reduced: trh_expr_create walks the program with a magnitude bound per stack entry (in units of m)
// and inserts a REDUCE (multiplication by the domain's one) wherever a sum could pass 250 m, a product 500 m (its top limb is
// value / 2^232: 512 m fills the signed 32-bit limb), the operand of a squaring 250 m or a stored value 15 m -- never for the
// reference's gate shapes.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "ctx.h"

struct trh_expr {
    int field;
    uint32_t n_insn, n_columns, n_outputs, n_consts, lds_slots;
    void* d_prog = nullptr;    // DevInsn[n_insn]
    void* d_consts = nullptr;  // n_consts x 32 B
    void* d_ptrs = nullptr;    // n_columns + n_outputs device pointers, refreshed per evaluation
    std::vector<const void*> h_ptrs;
    std::mutex mu;  // an evaluation stages the column pointers in the handle: one evaluation at a time, whichever context calls
};

namespace trh {
namespace {

typedef uint8_t u8;
typedef uint16_t u16;

struct DevInsn {
    u8 op, slot;  // slot: LDS slot spilled to (pushes) / refilled from (pops), 0xFF = none
    u16 a;
    i32 rot;
};
static_assert(sizeof(DevInsn) == 8, "instruction encoding");
// the kernel reads an instruction as two aligned dwords and unpacks the fields itself (hipcc 7.2 otherwise merges the
// u16 / i32 reads into one s_load_dword at byte offset 2, whose low address bits the hardware ignores)
struct RawInsn {
    u32 w0;  // op | slot << 8 | a << 16
    i32 rot;
};
constexpr u8 NO_SLOT = 0xFF;
constexpr int THREADS = 256;

constexpr u8 OP_REDUCE_TOP = 200, OP_REDUCE_NEXT = 201;  // internal: inserted by trh_expr_create's bound analysis

template <class F>
__device__ __forceinline__ void lds_put(unsigned char* smem, u32 slot, const Fy<F>& v) {
    uint4* a = (uint4*)(smem + (size_t)slot * (THREADS * 36));
    uint4* b = a + THREADS;
    u32* c = (u32*)(b + THREADS);
    a[threadIdx.x] = make_uint4((u32)v.l[0], (u32)v.l[1], (u32)v.l[2], (u32)v.l[3]);
    b[threadIdx.x] = make_uint4((u32)v.l[4], (u32)v.l[5], (u32)v.l[6], (u32)v.l[7]);
    c[threadIdx.x] = (u32)v.l[8];
}
template <class F>
__device__ __forceinline__ Fy<F> lds_get(const unsigned char* smem, u32 slot) {
    const uint4* a = (const uint4*)(smem + (size_t)slot * (THREADS * 36));
    const uint4* b = a + THREADS;
    const u32* c = (const u32*)(b + THREADS);
    const uint4 x = a[threadIdx.x], y = b[threadIdx.x];
    Fy<F> v;
    v.l[0] = (i32)x.x; v.l[1] = (i32)x.y; v.l[2] = (i32)x.z; v.l[3] = (i32)x.w;
    v.l[4] = (i32)y.x; v.l[5] = (i32)y.y; v.l[6] = (i32)y.z; v.l[7] = (i32)y.w;
    v.l[8] = (i32)c[threadIdx.x];
    return v;
}
// a constant of the program: stored by the host as the words of (c 2^261 mod m), non-negative and below m
template <class F>
__device__ __forceinline__ Fy<F> ldg_const(const uint4* p) {
    const uint4 a = p[0], b = p[1];
    return fy_load<F>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
// a column element: canonical Montgomery words (x 2^256 mod m) -> the domain value 32 (x 2^256) = x 2^261 + k m, 0 <= value < 32 m:
// limb k is the 29-bit window at bit 29 k - 5 of the words (the unpacking of fy_load, five bits lower)
template <class F>
__device__ __forceinline__ Fy<F> ldg_column(const uint4* p) {
    const uint4 a = p[0], b = p[1];
    const u32 M = (u32)YMASK;
    Fy<F> r;
    r.l[0] = (i32)((a.x << 5) & M);
    r.l[1] = (i32)(((a.x >> 24) | (a.y << 8)) & M);
    r.l[2] = (i32)(((a.y >> 21) | (a.z << 11)) & M);
    r.l[3] = (i32)(((a.z >> 18) | (a.w << 14)) & M);
    r.l[4] = (i32)(((a.w >> 15) | (b.x << 17)) & M);
    r.l[5] = (i32)(((b.x >> 12) | (b.y << 20)) & M);
    r.l[6] = (i32)(((b.y >> 9) | (b.z << 23)) & M);
    r.l[7] = (i32)(((b.z >> 6) | (b.w << 26)) & M);
    r.l[8] = (i32)(b.w >> 3);
    return r;
}

template <class F>
__global__ void __launch_bounds__(THREADS) expr_eval_kernel(const DevInsn* __restrict__ prog, u32 n_insn, const uint4* const* __restrict__ ptrs, u32 n_columns,
                                                            const uint4* __restrict__ consts, u32 log_n, u32 rot_step, u32 n_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // rows: one cyclic domain of 2^log_n rows (n_blocks = 0), or n_blocks cosets of 2^log_n rows each -- a rotation stays inside its block
    const size_t N = (size_t)1 << log_n, total = n_blocks ? (size_t)n_blocks << log_n : N;
    const size_t gid = (size_t)blockIdx.x * THREADS + threadIdx.x;
    const size_t row = gid < total ? gid : gid & (N - 1);  // surplus lanes of the last workgroup repeat rows, their stores are masked
    const bool live = gid < total;
    const size_t row_base = row & ~(N - 1);
    Fy<F> T = fy_zero<F>(), Nx = fy_zero<F>(), ACC = fy_zero<F>();
    for (u32 pc = 0; pc < n_insn; ++pc) {
        const RawInsn raw = ((const RawInsn*)prog)[pc];  // uniform: scalar loads
        DevInsn in;
        in.op = (u8)(raw.w0 & 0xffu); in.slot = (u8)((raw.w0 >> 8) & 0xffu); in.a = (u16)(raw.w0 >> 16); in.rot = raw.rot;
        switch (in.op) {
            case TRH_EXPR_PUSH_COLUMN: {
                if (in.slot != NO_SLOT) lds_put<F>(smem, in.slot, Nx);
                Nx = T;
                const size_t r = row_base | ((row + (size_t)((long long)in.rot * (long long)rot_step)) & (N - 1));
                T = ldg_column<F>(ptrs[in.a] + 2 * r);
                break;
            }
            case TRH_EXPR_PUSH_CONST:
                if (in.slot != NO_SLOT) lds_put<F>(smem, in.slot, Nx);
                Nx = T;
                T = ldg_const<F>(consts + 2 * (size_t)in.a);
                break;
            case TRH_EXPR_PUSH_LOCAL:
                if (in.slot != NO_SLOT) lds_put<F>(smem, in.slot, Nx);
                Nx = T;
                T = lds_get<F>(smem, in.a);
                break;
            case TRH_EXPR_ADD: T = fy_add(Nx, T); if (in.slot != NO_SLOT) Nx = lds_get<F>(smem, in.slot); break;
            case TRH_EXPR_SUB: T = fy_sub(Nx, T); if (in.slot != NO_SLOT) Nx = lds_get<F>(smem, in.slot); break;
            case TRH_EXPR_MUL: T = fy_mul(Nx, T); if (in.slot != NO_SLOT) Nx = lds_get<F>(smem, in.slot); break;
            case TRH_EXPR_NEG: T = fy_sub(fy_zero<F>(), T); break;
            case TRH_EXPR_SQR: T = fy_sqr(T); break;
            case TRH_EXPR_MUL_CONST: T = fy_mul(T, ldg_const<F>(consts + 2 * (size_t)in.a)); break;
            case TRH_EXPR_ADD_CONST: T = fy_add(T, ldg_const<F>(consts + 2 * (size_t)in.a)); break;
            case TRH_EXPR_STORE_LOCAL: lds_put<F>(smem, in.a, T); break;  // keeps T
            case TRH_EXPR_FOLD:  // acc = acc * const + T; pop
                ACC = fy_add(fy_mul(ACC, ldg_const<F>(consts + 2 * (size_t)in.a)), T);
                T = Nx;
                if (in.slot != NO_SLOT) Nx = lds_get<F>(smem, in.slot);
                break;
            case OP_REDUCE_TOP: if (in.a) ACC = fy_mul(ACC, fy_one<F>()); else T = fy_mul(T, fy_one<F>()); break;  // same value, |.| back below 3 m
            case OP_REDUCE_NEXT: Nx = fy_mul(Nx, fy_one<F>()); break;
            case TRH_EXPR_STORE_TOP:  // out[a][row] = T; pop
            case TRH_EXPR_STORE_ACC: {
                const Fe<F> v = fy_to_fe(in.op == TRH_EXPR_STORE_ACC ? ACC : T);
                if (live) {
                    u32 w[8];
                    fe_store(v, w);
                    uint4* o = (uint4*)ptrs[n_columns + in.a] + 2 * row;
                    o[0] = make_uint4(w[0], w[1], w[2], w[3]);
                    o[1] = make_uint4(w[4], w[5], w[6], w[7]);
                }
                if (in.op == TRH_EXPR_STORE_TOP) {
                    T = Nx;
                    if (in.slot != NO_SLOT) Nx = lds_get<F>(smem, in.slot);
                }
                break;
            }
            default: break;
        }
    }
}

// c (canonical Montgomery words, < m) -> 32 c mod m, the same words a column load produces lazily: five modular doublings
template <class F>
void const_to_domain(const uint64_t in[4], uint64_t out[4]) {
    u64 m[4], v[4];
    for (int i = 0; i < 4; ++i) { m[i] = (u64)F::MOD[2 * i] | ((u64)F::MOD[2 * i + 1] << 32); v[i] = in[i]; }
    auto geq = [&](const u64* a) { for (int i = 3; i >= 0; --i) { if (a[i] != m[i]) return a[i] > m[i]; } return true; };
    auto sub = [&](u64* a) {
        u64 br = 0;
        for (int i = 0; i < 4; ++i) {
            const u64 t = a[i] - m[i], b1 = a[i] < m[i] ? 1u : 0u, t2 = t - br, b2 = t < br ? 1u : 0u;
            a[i] = t2; br = b1 | b2;
        }
    };
    while (geq(v)) sub(v);  // tolerate a non-canonical input below 2^256
    for (int k = 0; k < 5; ++k) {
        const u64 top = v[3] >> 63;
        for (int i = 3; i > 0; --i) v[i] = (v[i] << 1) | (v[i - 1] >> 63);
        v[0] <<= 1;
        (void)top;  // m < 2^255 and v < m: 2 v < 2^256, no bit is lost
        if (geq(v)) sub(v);
    }
    for (int i = 0; i < 4; ++i) out[i] = v[i];
}

}  // namespace
}  // namespace trh

using namespace trh;

extern "C" {

int trh_expr_create(int field, const trh_expr_insn_t* insns, size_t n_insn, const uint64_t* consts, size_t n_consts, size_t n_columns, size_t n_outputs,
                    size_t n_locals, trh_expr** out) {
    TRH_TRY(require_init());
    if (field != TRH_FP && field != TRH_FQ) { set_error("unknown field id %d", field); return TRH_EINVAL; }
    if (!out || !insns || !n_insn || (n_consts && !consts)) { set_error("expr_create: null pointer / empty program"); return TRH_EINVAL; }
    if (n_columns > 65535 || n_outputs == 0 || n_outputs > 65535 || n_consts > 65535 || n_locals > 64) { set_error("expr_create: table sizes out of range"); return TRH_EINVAL; }
    // pass 1: stack depth at every instruction, maximum depth (the two top entries live in registers), and a magnitude bound (in
    // units of m) for every stack entry, local and the accumulator: the machine's values are lazy (expr.hip header); a REDUCE
    // goes in front of an instruction whose result could pass MAXV or whose stored value could pass what fy_to_fe accepts
    constexpr double MAXV = 250.0, MAX_STORE = 15.0, LOADED = 32.0, MAX_OPERAND = 500.0, MAX_SQR = 250.0;
    auto after_mul = [](double a, double b) { return a * b / 128.0 + 1.0; };  // |a b| / 2^261 + m, m / 2^261 < 2^-7
    std::vector<DevInsn> dev;
    dev.reserve(n_insn + 16);
    std::vector<double> bound;  // evaluation stack
    std::vector<double> local_bound(n_locals, 0.0);
    double acc_bound = 0.0;
    int depth = 0, max_depth = 0;
    auto reduce_top = [&]() { dev.push_back(DevInsn{OP_REDUCE_TOP, NO_SLOT, 0, 0}); bound.back() = after_mul(bound.back(), 1.0); };
    auto reduce_next = [&]() { dev.push_back(DevInsn{OP_REDUCE_NEXT, NO_SLOT, 0, 0}); bound[bound.size() - 2] = after_mul(bound[bound.size() - 2], 1.0); };
    auto reduce_acc = [&]() { dev.push_back(DevInsn{OP_REDUCE_TOP, NO_SLOT, 1, 0}); acc_bound = after_mul(acc_bound, 1.0); };
    for (size_t pc = 0; pc < n_insn; ++pc) {
        const trh_expr_insn_t& u = insns[pc];
        DevInsn d{(u8)u.op, NO_SLOT, (u16)u.a, u.rotation};
        auto bad = [&](const char* what) { set_error("expr_create: instruction %zu: %s", pc, what); return TRH_EINVAL; };
        switch (u.op) {
            case TRH_EXPR_PUSH_COLUMN: if (u.a >= n_columns) return bad("column index out of range"); bound.push_back(LOADED); goto push;
            case TRH_EXPR_PUSH_CONST: if (u.a >= n_consts) return bad("constant index out of range"); bound.push_back(1.0); goto push;
            case TRH_EXPR_PUSH_LOCAL: if (u.a >= n_locals) return bad("local index out of range"); bound.push_back(local_bound[u.a]);
            push:
                if (depth >= 2) d.slot = (u8)(depth - 2);  // the old next-of-stack is spilled below the two register entries
                ++depth;
                break;
            case TRH_EXPR_ADD: case TRH_EXPR_SUB: case TRH_EXPR_MUL: {
                if (depth < 2) return bad("binary operator needs two stack entries");
                if (u.op != TRH_EXPR_MUL && bound[bound.size() - 1] + bound[bound.size() - 2] > MAXV) {
                    if (bound[bound.size() - 1] >= bound[bound.size() - 2]) reduce_top(); else reduce_next();
                    if (bound[bound.size() - 1] + bound[bound.size() - 2] > MAXV) { if (bound[bound.size() - 1] >= bound[bound.size() - 2]) reduce_top(); else reduce_next(); }
                }
                // a product B m has the top limb B 2^22: operands and result have to stay below 512 m to fit the signed 32-bit limb.
                // Every value on the stack is <= MAX_OPERAND by construction (sums <= MAXV, products checked here), so one reduction of
                // the larger operand (-> below 5 m) is enough
                if (u.op == TRH_EXPR_MUL && after_mul(bound[bound.size() - 1], bound[bound.size() - 2]) > MAX_OPERAND) {
                    if (bound[bound.size() - 1] >= bound[bound.size() - 2]) reduce_top(); else reduce_next();
                }
                const double a = bound[bound.size() - 2], b = bound.back();
                bound.pop_back();
                bound.back() = u.op == TRH_EXPR_MUL ? after_mul(a, b) : a + b;
                --depth;
                if (depth >= 2) d.slot = (u8)(depth - 2);
                break;
            }
            case TRH_EXPR_NEG: case TRH_EXPR_SQR:
                if (depth < 1) return bad("unary operator on an empty stack");
                if (u.op == TRH_EXPR_SQR) {  // fy_squares doubles the limbs of its operand: below 256 m, and the result below 512 m
                    if (bound.back() > MAX_SQR) reduce_top();
                    bound.back() = after_mul(bound.back(), bound.back());
                }
                break;
            case TRH_EXPR_MUL_CONST: case TRH_EXPR_ADD_CONST:
                if (depth < 1) return bad("operator on an empty stack");
                if (u.a >= n_consts) return bad("constant index out of range");
                if (u.op == TRH_EXPR_ADD_CONST && bound.back() + 1.0 > MAXV) reduce_top();
                bound.back() = u.op == TRH_EXPR_MUL_CONST ? after_mul(bound.back(), 1.0) : bound.back() + 1.0;
                break;
            case TRH_EXPR_STORE_LOCAL:
                if (depth < 1) return bad("store of an empty stack");
                if (u.a >= n_locals) return bad("local index out of range");
                local_bound[u.a] = bound.back();
                break;
            case TRH_EXPR_FOLD:
                if (u.a >= n_consts) return bad("constant index out of range");
                if (depth < 1) return bad("pop of an empty stack");
                if (after_mul(acc_bound, 1.0) + bound.back() > MAXV) reduce_top();
                acc_bound = after_mul(acc_bound, 1.0) + bound.back();
                goto pop;
            case TRH_EXPR_STORE_TOP:
                if (u.a >= n_outputs) return bad("output index out of range");
                if (depth < 1) return bad("pop of an empty stack");
                if (bound.back() > MAX_STORE) reduce_top();
            pop:
                bound.pop_back();
                --depth;
                if (depth >= 2) d.slot = (u8)(depth - 2);
                break;
            case TRH_EXPR_STORE_ACC:
                if (u.a >= n_outputs) return bad("output index out of range");
                if (acc_bound > MAX_STORE) reduce_acc();
                break;
            default: return bad("unknown opcode");
        }
        if (depth > max_depth) max_depth = depth;
        if (max_depth - 2 > 120) return bad("evaluation stack deeper than 122 entries");
        dev.push_back(d);
    }
    const size_t n_dev = dev.size();
    const uint32_t stack_slots = max_depth > 2 ? (uint32_t)(max_depth - 2) : 0;
    // locals live behind the stack region
    for (size_t pc = 0; pc < n_dev; ++pc)
        if (dev[pc].op == TRH_EXPR_PUSH_LOCAL || dev[pc].op == TRH_EXPR_STORE_LOCAL) dev[pc].a = (u16)(dev[pc].a + stack_slots);
    const uint32_t slots = stack_slots + (uint32_t)n_locals;
    if ((size_t)slots * THREADS * 36 > 160 * 1024) { set_error("expr_create: %u LDS slots (stack %u + locals %zu) exceed the 160 KiB of a CU", slots, stack_slots, n_locals); return TRH_EINVAL; }

    trh_expr* e = new trh_expr();
    e->field = field; e->n_insn = (uint32_t)n_dev; e->n_columns = (uint32_t)n_columns; e->n_outputs = (uint32_t)n_outputs;
    e->n_consts = (uint32_t)n_consts; e->lds_slots = slots;
    e->h_ptrs.resize(n_columns + n_outputs);
    TRH_ENTER(0);
    Range range("trh_expr_create");
    Ctx& c = ctx();
    (void)c;
    hipError_t err = hipMalloc(&e->d_prog, n_dev * sizeof(DevInsn));
    if (err == hipSuccess) err = hipMalloc(&e->d_consts, (n_consts ? n_consts : 1) * 32);
    if (err == hipSuccess) err = hipMalloc(&e->d_ptrs, (n_columns + n_outputs) * sizeof(void*));
    if (err == hipSuccess) err = hipMemcpy(e->d_prog, dev.data(), n_dev * sizeof(DevInsn), hipMemcpyHostToDevice);
    std::vector<uint64_t> dom(4 * (n_consts ? n_consts : 1));
    for (size_t i = 0; i < n_consts; ++i) {
        if (field == TRH_FP) const_to_domain<FpParams>(consts + 4 * i, dom.data() + 4 * i);
        else const_to_domain<FqParams>(consts + 4 * i, dom.data() + 4 * i);
    }
    if (err == hipSuccess && n_consts) err = hipMemcpy(e->d_consts, dom.data(), n_consts * 32, hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        set_error("expr_create: %s", hipGetErrorString(err));
        if (e->d_prog) (void)hipFree(e->d_prog);
        if (e->d_consts) (void)hipFree(e->d_consts);
        if (e->d_ptrs) (void)hipFree(e->d_ptrs);
        delete e;
        return TRH_EHIP;
    }
    if (!(ctx().attr_done & ATTR_EXPR)) {  // per device
        TRH_HIP_TRY(hipFuncSetAttribute((const void*)expr_eval_kernel<FpParams>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TRH_HIP_TRY(hipFuncSetAttribute((const void*)expr_eval_kernel<FqParams>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx().attr_done |= ATTR_EXPR;
    }
    *out = e;
    return TRH_OK;
}

void trh_expr_destroy(trh_expr* e) {
    if (!e) return;
    if (e->d_prog) (void)hipFree(e->d_prog);
    if (e->d_consts) (void)hipFree(e->d_consts);
    if (e->d_ptrs) (void)hipFree(e->d_ptrs);
    delete e;
}

uint32_t trh_expr_lds_slots(trh_expr* e) { return e ? e->lds_slots : 0; }

/* replace one constant (the per-proof challenges y, beta, gamma, ... of an otherwise fixed program) */
int trh_expr_set_const(trh_expr* e, uint32_t index, const uint64_t value[4]) {
    TRH_TRY(require_init());
    if (!e || !value || index >= e->n_consts) { set_error("expr_set_const: bad arguments"); return TRH_EINVAL; }
    TRH_ENTER(0);
    Range range("trh_expr_set_const");
    Ctx& c = ctx();
    (void)c;
    uint64_t dom[4];
    if (e->field == TRH_FP) const_to_domain<FpParams>(value, dom); else const_to_domain<FqParams>(value, dom);
    TRH_HIP_TRY(hipMemcpy((char*)e->d_consts + (size_t)index * 32, dom, 32, hipMemcpyHostToDevice));
    return TRH_OK;
}

static int expr_eval(trh_expr* e, const void* const* columns_dev, void* const* outputs_dev, uint32_t log_n, uint32_t rot_step, uint32_t n_blocks, void* stream);
int trh_expr_eval_dev(trh_expr* e, const void* const* columns_dev, void* const* outputs_dev, uint32_t log_n, uint32_t rot_step, void* stream) {
    return expr_eval(e, columns_dev, outputs_dev, log_n, rot_step, 0, stream);
}
int trh_expr_eval_blocks_dev(trh_expr* e, const void* const* columns_dev, void* const* outputs_dev, uint32_t block_log, uint32_t n_blocks, void* stream) {
    if (n_blocks == 0 || n_blocks > 64) { set_error("expr_eval_blocks: n_blocks out of range"); return TRH_EINVAL; }
    return expr_eval(e, columns_dev, outputs_dev, block_log, 1, n_blocks, stream);
}
static int expr_eval(trh_expr* e, const void* const* columns_dev, void* const* outputs_dev, uint32_t log_n, uint32_t rot_step, uint32_t n_blocks, void* stream) {
    TRH_TRY(require_init());
    if (!e || !outputs_dev || (e->n_columns && !columns_dev)) { set_error("expr_eval: null pointer"); return TRH_EINVAL; }
    if (log_n > 30) { set_error("expr_eval: log_n %u too large", log_n); return TRH_EINVAL; }
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t i = 0; i < e->n_columns; ++i) {
        if (!columns_dev[i]) { set_error("expr_eval: column %u is null", i); return TRH_EINVAL; }
        e->h_ptrs[i] = columns_dev[i];
    }
    for (uint32_t i = 0; i < e->n_outputs; ++i) {
        if (!outputs_dev[i]) { set_error("expr_eval: output %u is null", i); return TRH_EINVAL; }
        e->h_ptrs[e->n_columns + i] = outputs_dev[i];
    }
    TRH_ENTER(stream);
    Range range("trh_expr_eval_dev");
    Ctx& c = ctx();
    (void)c;
    hipStream_t s = (hipStream_t)stream;
    TRH_HIP_TRY(hipMemcpyAsync(e->d_ptrs, e->h_ptrs.data(), e->h_ptrs.size() * sizeof(void*), hipMemcpyHostToDevice, s));
    const size_t N = n_blocks ? (size_t)n_blocks << log_n : (size_t)1 << log_n;
    const unsigned blocks = (unsigned)((N + THREADS - 1) / THREADS);
    const size_t lds = (size_t)e->lds_slots * THREADS * 36;
    if (e->field == TRH_FP)
        hipLaunchKernelGGL((expr_eval_kernel<FpParams>), dim3(blocks), dim3(THREADS), lds, s, (const DevInsn*)e->d_prog, e->n_insn, (const uint4* const*)e->d_ptrs, e->n_columns,
                           (const uint4*)e->d_consts, log_n, rot_step, n_blocks);
    else
        hipLaunchKernelGGL((expr_eval_kernel<FqParams>), dim3(blocks), dim3(THREADS), lds, s, (const DevInsn*)e->d_prog, e->n_insn, (const uint4* const*)e->d_ptrs, e->n_columns,
                           (const uint4*)e->d_consts, log_n, rot_step, n_blocks);
    TRH_HIP_TRY(hipGetLastError());
    TRH_HIP_TRY(hipStreamSynchronize(s));  // h_ptrs is reused by the next call
    return TRH_OK;
}

}  // extern "C"
