"""The reference's instruction chips, `unchanged`, signed-word and Mem gates (the 12 `create_gate` sites outside exe.rs / logic.rs /
even_bits.rs / sprod.rs) through the h(X) path, from the committed fixture tests/golden/chip_gates.json (tests/golden/make_chip_gates.py).

CPU part: the fixture is what the generator produces; a witness built from each chip's MEANING (TinyRAM semantics: add with carry,
product split into high and low words, signed sum, shifts through a power, the flag rules, registers that only change when flagged,
a memory trace sorted by address then time) satisfies every gate under the oracle's Expression::evaluate restatement, rows where no
chip is selected hold junk, and corrupted cells violate exactly their rows.
GPU part: `compile_gates` + the device evaluator agree with the oracle on every row for the satisfying and a corrupted witness."""
import importlib.util
import json
import os
import random
import sys

import numpy as np
import pytest

import pasta as o
from common import expr_from_json as from_json, expr_to_tuple as to_tuple
from tiny_ram_halo2_amd import expr

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "chip_gates.json")
FIELD = "fp"
K = 7


def load_fixture():
    with open(FIXTURE) as fh:
        return json.load(fh)


def even_odd(v, bits):
    """EvenBitsConfig decomposition: v = even + 2 odd, both with bits at even positions only"""
    e = od = 0
    for b in range(0, bits, 2):
        e |= ((v >> b) & 1) << b
        od |= ((v >> (b + 1)) & 1) << b
    assert e + 2 * od == v
    return e, od


def make_witness(doc, n, table_len, trace_len, seed):
    f = o.FIELDS[FIELD]
    rnd = random.Random(seed)
    R, WB = doc["reg_count"], doc["word_bits"]
    M = 1 << WB
    word = lambda: rnd.randrange(M)  # noqa: E731
    cols = {name: [rnd.randrange(1 << 40) for _ in range(n)] for name in doc["advice"]}  # junk everywhere first
    chips = ["flag1", "flag2", "flag3", "flag4", "mod", "prod", "ssum", "sum", "shift"]
    for r in range(n):
        cols["s_trace"][r] = 1 if r < trace_len else 0
        cols["m_s_trace"][r] = 1 if r < trace_len else 0
    for r in range(table_len):
        for c in chips:
            cols[f"s_{c}"][r] = 0
        cols["s_signed"][r] = 0
    # the last row wraps around to row 0 under Rotation::next: keep its `next` queries inside the junk-tolerant part (s_table = 0 there)
    for r in range(trace_len):
        nxt = r + 1
        chip = chips[r % len(chips)] if r % 11 != 10 else None  # some trace rows with no chip at all
        if chip:
            cols[f"s_{chip}"][r] = 1
        if chip == "sum":       # a + b = c + 2^W flag' - d, d = 0: addition with carry
            a, b = word(), word()
            cols["a"][r], cols["b"][r], cols["d"][r] = a, b, 0
            cols["c"][r], cols["flag"][nxt] = (a + b) % M, (a + b) // M
        elif chip == "prod":    # a b = d + 2^W c
            a, b = word(), word()
            cols["a"][r], cols["b"][r] = a, b
            cols["c"][r], cols["d"][r] = (a * b) // M, (a * b) % M
        elif chip == "mod":     # flag' (b - d) + d - b c - a = 0: division with remainder (flag' = 0), or by zero (flag' = 1, b = 0, a = 0)
            if r % 2:
                b, c = rnd.randrange(1, M), word()
                rem = rnd.randrange(b)
                cols["b"][r], cols["c"][r], cols["a"][r], cols["flag"][nxt] = b, c, (rem - b * c) % f.m, 0
                cols["d"][r] = rem
            else:
                cols["b"][r], cols["a"][r], cols["flag"][nxt] = 0, 0, 1   # d and c free
        elif chip == "flag1":   # flag' c = 0
            c = word() if r % 2 else 0
            cols["c"][r], cols["flag"][nxt] = c, (0 if c else 1)
        elif chip == "flag2":   # (flag' + c) a_flag = 1
            c, fl = word() | 1, rnd.randrange(2)
            cols["c"][r], cols["flag"][nxt] = c, fl
            cols["a_flag"][r] = pow(fl + c, -1, f.m)
        elif chip == "flag3":   # comparison: flag' = 0 and c - a - 1 = r = r_even + 2 r_odd, or flag' = 1 and b = 0 with r still c - a - 1
            a = word()
            rr = rnd.randrange(M - a - 1) if a < M - 1 else 0
            c = a + 1 + rr
            re, ro = even_odd(rr, WB)
            cols["a"][r], cols["c"][r], cols["r_word"][r], cols["r_even"][r], cols["r_odd"][r] = a, c, rr, re, ro
            if r % 2:
                cols["flag"][nxt] = 0
            else:
                cols["flag"][nxt], cols["b"][r] = 1, 0
        elif chip == "flag4":   # flag' = b_flag msb_b + (1 - b_flag) lsb_b
            bf, ms, ls = rnd.randrange(2), rnd.randrange(2), rnd.randrange(2)
            cols["b_flag"][r], cols["msb_b"][r], cols["lsb_b"][r] = bf, ms, ls
            cols["flag"][nxt] = bf * ms + (1 - bf) * ls
        elif chip == "ssum":    # signed a + b = signed c + 2^W flag' - d with word_sigma = magnitude, msb = sign
            am, cm, asg, csg, b, fl = word(), word(), rnd.randrange(2), rnd.randrange(2), word(), rnd.randrange(2)
            cols["a_sigma"][r], cols["a_msb"][r], cols["c_sigma"][r], cols["c_msb"][r], cols["b"][r], cols["flag"][nxt] = am, asg, cm, csg, b, fl
            sa, sc = (-am if asg else am), (-cm if csg else cm)
            cols["d"][r] = (sc + M * fl - sa - b) % f.m
        elif chip == "shift":   # a_shift boolean; W - a = rs_even + 2 rs_odd when not saturated; a_power b = d + 2^W c
            a = rnd.randrange(WB + 1)
            re, ro = even_odd(WB - a, WB)
            b = word()
            cols["a"][r], cols["b"][r], cols["a_shift"][r], cols["rs_even"][r], cols["rs_odd"][r] = a, b, 0, re, ro
            cols["a_power"][r] = 1 << a
            cols["c"][r], cols["d"][r] = ((b << a) // M), ((b << a) % M)
            if r % 4 == 0:      # saturated shift (a > W in the instruction): a_shift = 1 lifts the range relation
                cols["a_shift"][r], cols["a"][r] = 1, word()
        # signed word on every third trace row
        if r % 3 == 0:
            cols["s_signed"][r] = 1
            w = word()
            msb = w >> (WB - 1)
            cols["sg_word"][r], cols["sg_msb"][r] = w, msb
            cols["sg_sigma"][r] = (M - w) if msb else w
            odd = word()
            cols["sg_odd"][r] = odd
            cols["sg_check"][r] = (odd + (1 - 2 * msb) * (1 << (WB - 2))) % f.m
    # unchanged: registers / pc / flag only move when their `changed` flag is set (rows r -> r + 1 inside the trace)
    for r in range(trace_len - 1):
        cols["ch_pc"][r] = rnd.randrange(2)
        if not cols["ch_pc"][r]:
            cols["pc"][r + 1] = cols["pc"][r] + 1
        for i in range(R):
            cols[f"ch_reg{i}"][r] = rnd.randrange(2)
            if not cols[f"ch_reg{i}"][r]:
                cols[f"reg{i}"][r + 1] = cols[f"reg{i}"][r]
    for r in range(trace_len - 1):  # the flag is written by the chips above: mark it changed wherever it differs
        cols["ch_flag"][r] = 0 if cols["flag"][r + 1] == cols["flag"][r] else 1
    # Mem: accesses sorted by address, then time
    addr, t, val = rnd.randrange(100), rnd.randrange(100), word()
    for r in range(trace_len):
        cols["address"][r], cols["time"][r], cols["value"][r] = addr, t, val
        cols["load"][r] = 0
        if r + 1 < trace_len:
            if rnd.randrange(3):    # same address: later time, not an initial value; a load keeps the value
                inc = rnd.randrange(1, 50)
                cols["time_inc"][r + 1], cols["init"][r + 1] = inc, 0
                t += inc
                if rnd.randrange(2):
                    cols["load"][r] = 1
                else:
                    val = word()
            else:                   # next access trace on a larger address
                gap = rnd.randrange(50)
                cols["addr_inc"][r + 1] = gap
                addr += 1 + gap
                t, val = rnd.randrange(100), word()
    ix = {name: i for i, name in enumerate(doc["advice"])}
    out = {("advice", ix[name]): [v % f.m for v in col] for name, col in cols.items()}
    out[("selector", doc["selectors"].index("s_table"))] = [1] * table_len + [0] * (n - table_len)
    out[("selector", doc["selectors"].index("m_s_table"))] = [1] * table_len + [0] * (n - table_len)
    return out, ix


def folded_rows(f, gates, cols, y, n):
    return o.evaluate_gates(f, [to_tuple(g) for g in gates], cols, y, n)


def test_fixture_is_what_the_generator_produces():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    spec = importlib.util.spec_from_file_location("make_chip_gates", os.path.join(HERE, "golden", "make_chip_gates.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    doc = load_fixture()
    built = mod.build_gates()
    assert [g["name"] for g in doc["gates"]] == [n for n, _ in built]
    assert [g["expr"] for g in doc["gates"]] == [json.loads(json.dumps(mod.to_json(g))) for _, g in built]
    names = [g["name"] for g in doc["gates"]]
    assert sorted(set(names)) == sorted(["unchanged", "flag1", "flag2", "flag3", "flag4", "mod", "prod", "ssum", "sum", "shift", "signed", "Mem"])
    assert names.count("unchanged") == 10 and names.count("Mem") == 4 and names.count("shift") == 3 and len(names) == 28


def test_witness_satisfies_every_chip_and_corruption_is_caught():
    f = o.FIELDS[FIELD]
    doc = load_fixture()
    gates = [from_json(g["expr"]) for g in doc["gates"]]
    n, table_len, trace_len = 1 << K, 100, 64
    cols, ix = make_witness(doc, n, table_len, trace_len, 0xC41F)
    y = 0x5EED5EED5EED5EED1234 % f.m
    assert not any(folded_rows(f, gates, cols, y, n))
    chips = ["flag1", "flag2", "flag3", "flag4", "mod", "prod", "ssum", "sum", "shift"]
    for chip, cell in (("sum", "c"), ("prod", "d"), ("shift", "a_power"), ("flag2", "a_flag"), ("ssum", "d"), ("mod", "a"), ("flag4", "lsb_b")):
        row = next(r for r in range(3, trace_len - 1) if cols[("advice", ix[f"s_{chip}"])][r] == 1 and (chip != "flag4" or cols[("advice", ix["b_flag"])][r] == 0)
                   and (chip != "mod" or r % 2))
        bad = {k: list(v) for k, v in cols.items()}
        bad[("advice", ix[cell])][row] = (bad[("advice", ix[cell])][row] + 1) % f.m
        assert [i for i, v in enumerate(folded_rows(f, gates, bad, y, n)) if v] == [row], (chip, cell)
    # a register that changes without its flag: the `unchanged` gate of the row before
    bad = {k: list(v) for k, v in cols.items()}
    r = next(r for r in range(2, trace_len - 2) if cols[("advice", ix["ch_reg3"])][r] == 0)
    bad[("advice", ix["reg3"])][r + 1] += 1
    hit = [i for i, v in enumerate(folded_rows(f, gates, bad, y, n)) if v]
    assert r in hit and set(hit) <= {r, r + 1}
    # memory: a load that returns another value than the one stored
    bad = {k: list(v) for k, v in cols.items()}
    r = next(r for r in range(trace_len - 1) if cols[("advice", ix["load"])][r] == 1)
    bad[("advice", ix["value"])][r + 1] += 1
    assert r in [i for i, v in enumerate(folded_rows(f, gates, bad, y, n)) if v]
    del chips


@pytest.mark.gpu
def test_chip_gates_on_the_device():
    import torch
    from tiny_ram_halo2_amd import api
    api.init(0)
    f = o.FIELDS[FIELD]
    doc = load_fixture()
    gates = [from_json(g["expr"]) for g in doc["gates"]]
    n, table_len, trace_len = 1 << K, 100, 64
    cols, ix = make_witness(doc, n, table_len, trace_len, 0xC41F)
    y = 0x5EED5EED5EED5EED1234 % f.m
    prog = expr.compile_gates(FIELD, gates, y)
    ev = expr.GateEvaluator(prog)

    def to_dev(col):
        return torch.from_numpy(np.array([f.limbs(v) for v in col], np.uint64).view(np.int64)).cuda()

    def from_dev(t):
        torch.cuda.synchronize()
        return [f.from_limbs(r) for r in t.contiguous().cpu().numpy().view(np.uint64).reshape(-1, 4)]

    used = set(prog.columns)
    assert not any(from_dev(ev.eval({k: to_dev(v) for k, v in cols.items() if k in used}, K, 1)))
    bad = {k: list(v) for k, v in cols.items()}
    for name, row in (("c", 7), ("a_power", 8), ("value", 20), ("reg5", 13), ("sg_sigma", 9)):
        bad[("advice", ix[name])][row] = (bad[("advice", ix[name])][row] + 3) % f.m
    got = from_dev(ev.eval({k: to_dev(v) for k, v in bad.items() if k in used}, K, 1))
    want = folded_rows(f, gates, bad, y, n)
    assert got == want and any(want)
