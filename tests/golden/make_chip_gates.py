#!/usr/bin/env python3
"""Generates tests/golden/chip_gates.json: every remaining `create_gate` of the reference's circuit -- the instruction chips the Exe
table wires to its temporary variables a, b, c, d and the flag, the `unchanged` gate, the signed-word gate and the Mem table's gate --
transcribed from

    changed.rs:91-120  flag1.rs:32-40  flag2.rs:38-50  flag3.rs:43-78  flag4.rs:40-55  modulo.rs:40-54  prod.rs:62-74
    ssum.rs:73-100     sum.rs:78-96    shift.rs:112-140  tables/signed.rs:65-106  tables/mem.rs:107-154
    (selector helpers: tables/mod.rs:36-54)

under /root/reference/src/circuits.  Together with tests/golden/exe_tempvar_gates.json (exe.rs, 12 sites), the logic chip (logic.rs, 4
sites), even_bits.rs and sprod.rs (tests/test_gpu_logic_chip.py) this covers all 30 `create_gate` sites of the reference.
The JSON holds expression trees over named columns; tests/test_chip_gates.py builds a witness from each chip's MEANING and checks it
under the oracle (CPU) and through `compile_gates` / the device evaluator (GPU).  WORD_BITS = 16, REG_COUNT = 8.

    python tests/golden/make_chip_gates.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tiny_ram_halo2_amd import expr  # noqa: E402
from make_exe_gates import to_json  # noqa: E402

REG_COUNT = 8
WORD_BITS = 16
MAX = 1 << WORD_BITS

ADVICE = (["s_trace", "a", "b", "c", "d", "flag", "pc"] + [f"reg{i}" for i in range(REG_COUNT)]
          + ["ch_pc", "ch_flag"] + [f"ch_reg{i}" for i in range(REG_COUNT)]
          + ["s_flag1", "s_flag2", "s_flag3", "s_flag4", "s_mod", "s_prod", "s_ssum", "s_sum", "s_shift", "s_signed"]
          + ["a_flag", "r_word", "r_even", "r_odd", "b_flag", "msb_b", "lsb_b", "a_sigma", "a_msb", "c_sigma", "c_msb",
             "a_shift", "a_power", "rs_even", "rs_odd", "sg_word", "sg_odd", "sg_msb", "sg_sigma", "sg_check"]
          + ["m_s_trace", "address", "time", "init", "load", "value", "addr_inc", "time_inc"])
IX = {n: i for i, n in enumerate(ADVICE)}
SELECTORS = ["s_table", "m_s_table"]


def adv(name, rot=0):
    return expr.Advice(IX[name], rot)


def sel(name):
    return expr.Selector(SELECTORS.index(name))


def const(v):
    return expr.Constant(v)


def with_selector(selector, constraints):
    return [selector * c for c in constraints]


def build_gates():
    one, two = const(1), const(2)
    g = []

    def add(name, selector, constraints):
        g.extend((name, e) for e in with_selector(selector, constraints))

    # changed.rs:91-120 "unchanged": s_extent = s_table * s_trace(next)
    cs = [(one - adv("ch_pc")) * (adv("pc") + one - adv("pc", 1)), (one - adv("ch_flag")) * (adv("flag") - adv("flag", 1))]
    cs += [(one - adv(f"ch_reg{i}")) * (adv(f"reg{i}") - adv(f"reg{i}", 1)) for i in range(REG_COUNT)]
    add("unchanged", sel("s_table") * adv("s_trace", 1), cs)
    # flag1.rs:32-40
    add("flag1", sel("s_table") * adv("s_flag1"), [adv("flag", 1) * adv("c")])
    # flag2.rs:38-50
    add("flag2", sel("s_table") * adv("s_flag2"), [(adv("flag", 1) + adv("c")) * adv("a_flag") - one])
    # flag3.rs:43-78
    flag_n = adv("flag", 1)
    add("flag3", sel("s_table") * adv("s_flag3"),
        [adv("b") * flag_n + (one - flag_n) * (adv("c") - adv("a") - one - two * adv("r_odd") - adv("r_even")),
         adv("c") * ((adv("c") - adv("a") - one) - adv("r_word"))])
    # flag4.rs:40-55
    add("flag4", sel("s_table") * adv("s_flag4"), [adv("flag", 1) - (adv("b_flag") * adv("msb_b")) - ((one - adv("b_flag")) * adv("lsb_b"))])
    # modulo.rs:40-54
    add("mod", sel("s_table") * adv("s_mod"), [adv("flag", 1) * (adv("b") - adv("d")) + adv("d") - adv("b") * adv("c") - adv("a")])
    # prod.rs:62-74
    add("prod", sel("s_table") * adv("s_prod"), [adv("a") * adv("b") - adv("d") - const(MAX) * adv("c")])
    # ssum.rs:73-100
    a_sigma = -adv("a_msb") * two * adv("a_sigma") + adv("a_sigma")
    c_sigma = -adv("c_msb") * two * adv("c_sigma") + adv("c_sigma")
    add("ssum", sel("s_table") * adv("s_ssum"), [a_sigma + adv("b") - c_sigma - (const(MAX) * adv("flag", 1)) + adv("d")])
    # sum.rs:78-96
    add("sum", sel("s_table") * adv("s_sum"), [adv("a") + adv("b") - adv("c") - (const(MAX) * adv("flag", 1)) + adv("d")])
    # shift.rs:112-140
    add("shift", sel("s_table") * adv("s_shift"),
        [adv("a_shift") * (adv("a_shift") - one),
         (one - adv("a_shift")) * (const(WORD_BITS) - adv("a") - (two * adv("rs_odd")) - adv("rs_even")),
         adv("a_power") * adv("b") - adv("d") - const(MAX) * adv("c")])
    # tables/signed.rs:65-106: s_signed(meta) = s_table * (sum of the selector columns), here one column
    word_sigma = -adv("sg_msb") * two * adv("sg_sigma") + adv("sg_sigma")
    add("signed", sel("s_table") * adv("s_signed"),
        [(-adv("sg_msb") * const(MAX) + adv("sg_word")) - word_sigma,
         adv("sg_odd") + (one - two * adv("sg_msb")) * const(1 << (WORD_BITS - 2)) - adv("sg_check")])
    # tables/mem.rs:107-154
    address_next, address = adv("address", 1), adv("address")
    same_cycle = address_next - address
    end_cycle = address_next - address - one - adv("addr_inc", 1)
    time_sorted = adv("time", 1) - adv("time") - adv("time_inc", 1)
    add("Mem", sel("m_s_table") * adv("m_s_trace", 1),
        [end_cycle * same_cycle, end_cycle * time_sorted, end_cycle * adv("init", 1), adv("load") * (adv("value", 1) - adv("value"))])
    return g


def main():
    gates = build_gates()
    doc = {"source": "/root/reference/src/circuits/{changed,flag1,flag2,flag3,flag4,modulo,prod,ssum,sum,shift}.rs, tables/{signed,mem}.rs (line ranges in the generator)",
           "generator": "tests/golden/make_chip_gates.py", "reg_count": REG_COUNT, "word_bits": WORD_BITS, "advice": ADVICE, "selectors": SELECTORS,
           "gates": [{"name": n, "degree": e.degree(), "expr": to_json(e)} for n, e in gates]}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chip_gates.json")
    with open(out, "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))
        fh.write("\n")
    print(f"{len(gates)} gate polynomials, max degree {max(e.degree() for _, e in gates)}, {len(ADVICE)} advice columns -> {out}")


if __name__ == "__main__":
    main()
