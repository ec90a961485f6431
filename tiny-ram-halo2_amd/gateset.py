"""The reference circuit's gate polynomials as ONE list for the h(X) timing run of the create_proof replay.

The 30 `create_gate` sites of /root/reference/src/circuits are already transcribed, as expression trees over named columns, in the
committed fixtures tests/golden/exe_tempvar_gates.json (exe.rs:147-498: 90 gate polynomials) and tests/golden/chip_gates.json (the
instruction chips, `unchanged`, the signed-word gate, the Mem table: 28); the logic chip (logic.rs:125-185), the even-bits decomposition
gate (even_bits.rs:143-157) and the degree-6 sprod gate (sprod.rs:65-92) are the three small identities written out below.  This
module merges them with the multiplicities the circuit instantiates them with: the signed-word gate three
times (signed_a / _b / _c, exe.rs:661-692), the even-bits gate 14 times (14 `EvenBitsConfig`s), everything else once; columns
that carry the same name in two fixtures (pc, flag, reg0..7, s_trace, value) are the same column, as in the Exe table.

What this is NOT: the assembled constraint system.  Which selector enables which chip on which row is the circuit's `configure` /
assignment code and stays on the Rust side; for the TIMING of the gate evaluator only the polynomials' shapes (degrees, the columns and
rotations they read) matter.  The fixtures are data files of the test tree: the caller names their directory (argument or TRH_GATES_DIR); `reference_gates()` returns
None when it is not given or the files are not there.
"""
from __future__ import annotations

import json
import os

from . import expr

GATES_DIR_ENV = "TRH_GATES_DIR"  # where the caller keeps the two fixture files; the package itself knows no path outside itself
WORD_BITS = 16
N_EVEN_BITS_CONFIGS = 14
N_SIGNED_CONFIGS = 3


class _Columns:
    """name -> column index, one numbering for the whole circuit"""

    def __init__(self):
        self.advice, self.selectors = {}, {}

    def adv(self, name, rot=0):
        return expr.Advice(self.advice.setdefault(name, len(self.advice)), rot)

    def sel(self, name):
        return expr.Selector(self.selectors.setdefault(name, len(self.selectors)))


def _from_json(j, names, sel_names, cols, rename):
    tag = j[0]
    if tag == "const":
        return expr.Constant(int(j[1], 16))
    if tag == "advice":
        return cols.adv(rename(names[j[1]]), j[2])
    if tag == "selector":
        return cols.sel(sel_names[j[1]])
    if tag == "neg":
        return expr.Negated(_from_json(j[1], names, sel_names, cols, rename))
    if tag == "sum":
        return expr.Sum(_from_json(j[1], names, sel_names, cols, rename), _from_json(j[2], names, sel_names, cols, rename))
    if tag == "prod":
        return expr.Product(_from_json(j[1], names, sel_names, cols, rename), _from_json(j[2], names, sel_names, cols, rename))
    if tag == "scaled":
        return expr.Scaled(_from_json(j[1], names, sel_names, cols, rename), int(j[2], 16))
    raise ValueError(tag)


def reference_gates(golden_dir: str | None = None):
    """-> (gates, info) or None.  gates: list of Expression over Advice / Selector columns numbered by `info["advice"]` / ["selectors"].
    golden_dir: the directory that holds exe_tempvar_gates.json and chip_gates.json (the repository keeps them under tests/golden/; the
    replay's callers pass it, or set TRH_GATES_DIR); without one there is no reference gate set and the replay times the synthetic one only"""
    gd = golden_dir or os.environ.get(GATES_DIR_ENV)
    if not gd:
        return None
    paths = [os.path.join(gd, "exe_tempvar_gates.json"), os.path.join(gd, "chip_gates.json")]
    if not all(os.path.exists(p) for p in paths):
        return None
    cols = _Columns()
    gates, by_site = [], {}

    def add(site, g):
        gates.append(g)
        by_site[site] = by_site.get(site, 0) + 1

    ident = lambda name: name  # noqa: E731
    with open(paths[0]) as fh:
        doc = json.load(fh)
    for g in doc["gates"]:
        add("exe.rs temp-var / trace gates", _from_json(g["expr"], doc["advice"], doc["selectors"], cols, ident))
    with open(paths[1]) as fh:
        doc = json.load(fh)
    for g in doc["gates"]:
        copies = N_SIGNED_CONFIGS if g["name"] == "signed" else 1
        for c in range(copies):  # signed_a / signed_b / signed_c own their sg_* columns
            ren = (lambda name, c=c: f"{name}.{c}" if name.startswith("sg_") and c else name)
            add(g["name"], _from_json(g["expr"], doc["advice"], doc["selectors"], cols, ren))
    s_table, two = cols.sel("s_table"), expr.Constant(2)
    # even-bits decomposition (even_bits.rs:143-157): s (even + 2 odd - word), s the enabling expression handed to configure
    instr = ["s_and", "s_xor", "s_or", "s_mod", "s_ssum", "s_sprod", "s_shift"]
    for i in range(N_EVEN_BITS_CONFIGS):
        s = s_table * cols.adv(instr[i % len(instr)])
        add("even_bits", s * (cols.adv(f"eb{i}_even") + two * cols.adv(f"eb{i}_odd") - cols.adv(f"eb{i}_word")))
    # logic chip (logic.rs:125-185): a, b, even_sum, odd_sum are four of the decompositions above
    s = s_table * (cols.adv("s_and") + cols.adv("s_xor") + cols.adv("s_or"))
    a_e, a_o, b_e, b_o = cols.adv("eb0_even"), cols.adv("eb0_odd"), cols.adv("eb1_even"), cols.adv("eb1_odd")
    es, os_ = cols.adv("eb2_word"), cols.adv("eb3_word")
    es_e, es_o, os_e, os_o = cols.adv("eb2_even"), cols.adv("eb2_odd"), cols.adv("eb3_even"), cols.adv("eb3_odd")
    res = cols.adv("c")
    add("logic", s * (a_e + b_e - es))
    add("logic", s * (a_o + b_o - os_))
    and_, xor = es_o + two * os_o, es_e + two * os_e
    add("logic", s_table * cols.adv("s_and") * (and_ - res))
    add("logic", s_table * cols.adv("s_xor") * (xor - res))
    add("logic", s_table * cols.adv("s_or") * (xor + and_ - res))
    # sprod (sprod.rs:65-92), degree 6: s_table s_sprod (a_s b_s - d - 2^W c_s), x_s = x_sigma (1 - 2 x_msb)
    signed = lambda v: -cols.adv(f"{v}_msb") * two * cols.adv(f"{v}_sigma") + cols.adv(f"{v}_sigma")  # noqa: E731
    add("sprod", s_table * cols.adv("s_sprod") * (signed("a") * signed("b") - cols.adv("d") - expr.Constant(1 << WORD_BITS) * signed("c")))
    info = {"gates": len(gates), "by_site": by_site, "advice": len(cols.advice), "selectors": len(cols.selectors),
            "max_degree": max(g.degree() for g in gates), "degree_histogram": {}}
    for g in gates:
        info["degree_histogram"][g.degree()] = info["degree_histogram"].get(g.degree(), 0) + 1
    return gates, info
