#!/usr/bin/env python3
"""Generates tests/golden/exe_tempvar_gates.json: the gate family of the reference's Exe table that ties the four temporary
variables a, b, c, d of a TinyRAM step to their sources, plus the two trace-extent gates -- transcribed gate by gate from
/root/reference/src/circuits/tables/exe.rs:147-498 (trace_len_gates :147-192; pc_gate :193-213, pc_gate_plus_one :215-234,
pc_next_gate :236-262, reg_gate :264-287, reg_next_gate :289-314, immediate_gate :316-335, vaddr_gate :337-357, one_gate
:359-377, zero_gate :379-394, max_word_gate :396-423; configure_selectors_a..d :425-498) with the selector helpers of
tables/mod.rs:36-54 (`query` = s_table * s_trace, `query_trace_next` = s_table * s_trace(next)) and Answer::OP_CODE = 0b11111
(instructions/opcode.rs:83).  REG_COUNT = 8 as in the reference's tests; WORD_BITS is a parameter of the fixture (max_word).

There is no Rust toolchain in this image, so the expressions cannot be dumped from halo2's ConstraintSystem; this script IS the
transcription, and the JSON is its output (data: expression trees over named columns).  tests/test_exe_gates.py checks the
fixture against the oracle's Expression::evaluate restatement on a satisfying witness (CPU) and runs it through
`compile_gates` / the device evaluator (GPU).

    python tests/golden/make_exe_gates.py        # rewrites the JSON next to this file
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tiny_ram_halo2_amd import expr  # noqa: E402  (the Python mirror of halo2's Expression<F>)

REG_COUNT = 8
WORD_BITS = 16
ANSWER_OP_CODE = 0b11111
R_CONST = (1 << 64) - 1  # u64::MAX, exe.rs:173


def advice_layout():
    names = ["s_trace", "pc", "flag"] + [f"reg{i}" for i in range(REG_COUNT)] + ["opcode", "immediate", "value", "tv_a", "tv_b", "tv_c", "tv_d"]
    names += ["sa_pc_next"] + [f"sa_reg{i}" for i in range(REG_COUNT)] + [f"sa_reg_next{i}" for i in range(REG_COUNT)] + ["sa_a", "sa_v_addr"]
    names += ["sb_pc", "sb_pc_next", "sb_pc_plus_one"] + [f"sb_reg{i}" for i in range(REG_COUNT)] + [f"sb_reg_next{i}" for i in range(REG_COUNT)] + ["sb_a", "sb_max_word"]
    names += [f"sc_reg{i}" for i in range(REG_COUNT)] + [f"sc_reg_next{i}" for i in range(REG_COUNT)] + ["sc_a", "sc_zero"]
    names += ["sd_pc"] + [f"sd_reg{i}" for i in range(REG_COUNT)] + [f"sd_reg_next{i}" for i in range(REG_COUNT)] + ["sd_a", "sd_zero", "sd_one"]
    return names


ADVICE = advice_layout()
IX = {n: i for i, n in enumerate(ADVICE)}
SELECTORS = ["first_line", "s_table"]


def adv(name, rot=0):
    return expr.Advice(IX[name], rot)


def sel(name):
    return expr.Selector(SELECTORS.index(name))


def one():
    return expr.Constant(1)


def with_selector(selector, constraints):  # Constraints::with_selector: selector * constraint for each
    return [selector * c for c in constraints]


def build_gates():
    gates = []  # (name, Expression)
    # trace_len_gates, exe.rs:147-192
    trace_starts = one() - adv("s_trace")
    cs = [trace_starts, adv("pc"), adv("flag")] + [adv(f"reg{i}") for i in range(REG_COUNT)]
    gates += [("start_trace", g) for g in with_selector(sel("first_line"), cs)]
    r = expr.Constant(R_CONST)
    contiguous = adv("s_trace") - adv("s_trace", 1)
    may_change = r - (adv("s_trace") * r) + adv("opcode") - expr.Constant(ANSWER_OP_CODE)
    gates += [("contiguous_trace", g) for g in with_selector(sel("s_table"), [contiguous * may_change])]

    def query():             # tables/mod.rs:36-43
        return sel("s_table") * adv("s_trace")

    def query_trace_next():  # tables/mod.rs:47-54
        return sel("s_table") * adv("s_trace", 1)

    def pc_gate(s, tv, name):  # exe.rs:193-213
        return [(f"tv.{name}.pc", g) for g in with_selector(sel("s_table") * adv("s_trace", 1) * adv(s), [adv("pc") - adv(tv)])]

    def pc_gate_plus_one(s, tv, name):  # :215-234
        return [(f"tv.{name}.pc+1", g) for g in with_selector(query_trace_next() * adv(s), [(adv("pc") + one()) - adv(tv)])]

    def pc_next_gate(s, tv, name):  # :236-262
        return [(f"tv.{name}.pc_next", g) for g in with_selector(query_trace_next() * adv(s), [adv("pc", 1) - adv(tv)])]

    def reg_gate(prefix, tv, name):  # :264-287
        return [(f"tv.{name}.reg[{i}]", g) for i in range(REG_COUNT) for g in with_selector(query() * adv(f"{prefix}_reg{i}"), [adv(f"reg{i}") - adv(tv)])]

    def reg_next_gate(prefix, tv, name):  # :289-314
        return [(f"tv.{name}.reg_next[{i}]", g) for i in range(REG_COUNT)
                for g in with_selector(query_trace_next() * adv(f"{prefix}_reg_next{i}"), [adv(f"reg{i}", 1) - adv(tv)])]

    def immediate_gate(s, tv, name):  # :316-335
        return [(f"tv.{name}.a", g) for g in with_selector(query() * adv(s), [adv("immediate") - adv(tv)])]

    def vaddr_gate(s, tv, name):  # :337-357
        return [(f"tv.{name}.vaddr", g) for g in with_selector(sel("s_table") * adv("s_trace") * adv(s), [adv("value") - adv(tv)])]

    def one_gate(s, tv, name):  # :359-377
        return [(f"tv.{name}.one", g) for g in with_selector(sel("s_table") * adv("s_trace") * adv(s), [one() - adv(tv)])]

    def zero_gate(s, tv, name):  # :379-394
        return [(f"tv.{name}.zero", g) for g in with_selector(sel("s_table") * adv("s_trace") * adv(s), [adv(tv)])]

    def max_word_gate(s, tv, name):  # :396-423
        return [(f"tv.{name}.max_word", g) for g in with_selector(sel("s_table") * adv("s_trace") * adv(s), [expr.Constant((1 << WORD_BITS) - 1) - adv(tv)])]

    # configure_selectors_a .. d, exe.rs:425-498
    gates += pc_next_gate("sa_pc_next", "tv_a", "a") + reg_gate("sa", "tv_a", "a") + reg_next_gate("sa", "tv_a", "a") + immediate_gate("sa_a", "tv_a", "a") + vaddr_gate("sa_v_addr", "tv_a", "a")
    gates += (pc_gate("sb_pc", "tv_b", "b") + pc_next_gate("sb_pc_next", "tv_b", "b") + pc_gate_plus_one("sb_pc_plus_one", "tv_b", "b") + reg_gate("sb", "tv_b", "b")
              + reg_next_gate("sb", "tv_b", "b") + immediate_gate("sb_a", "tv_b", "b") + max_word_gate("sb_max_word", "tv_b", "b"))
    gates += reg_gate("sc", "tv_c", "c") + reg_next_gate("sc", "tv_c", "c") + immediate_gate("sc_a", "tv_c", "c") + zero_gate("sc_zero", "tv_c", "c")
    gates += (pc_gate("sd_pc", "tv_d", "d") + reg_gate("sd", "tv_d", "d") + reg_next_gate("sd", "tv_d", "d") + immediate_gate("sd_a", "tv_d", "d")
              + zero_gate("sd_zero", "tv_d", "d") + one_gate("sd_one", "tv_d", "d"))
    return gates


def to_json(e):
    if isinstance(e, expr.Constant):
        return ["const", hex(e.value)]
    if isinstance(e, expr._Query):
        return [e.kind, e.column, e.rotation]
    if isinstance(e, expr.Negated):
        return ["neg", to_json(e.e)]
    if isinstance(e, expr.Sum):
        return ["sum", to_json(e.a), to_json(e.b)]
    if isinstance(e, expr.Product):
        return ["prod", to_json(e.a), to_json(e.b)]
    if isinstance(e, expr.Scaled):
        return ["scaled", to_json(e.e), hex(e.value)]
    raise TypeError(e)


def main():
    gates = build_gates()
    doc = {"source": "/root/reference/src/circuits/tables/exe.rs:147-498 (+ tables/mod.rs:36-54, instructions/opcode.rs:83)",
           "generator": "tests/golden/make_exe_gates.py", "reg_count": REG_COUNT, "word_bits": WORD_BITS,
           "advice": ADVICE, "selectors": SELECTORS,
           "gates": [{"name": n, "degree": g.degree(), "expr": to_json(g)} for n, g in gates]}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exe_tempvar_gates.json")
    with open(out, "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))
        fh.write("\n")
    print(f"{len(gates)} gate polynomials, max degree {max(g.degree() for _, g in gates)}, {len(ADVICE)} advice columns -> {out}")


if __name__ == "__main__":
    main()
