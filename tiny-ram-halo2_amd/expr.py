"""Host-side mirror of halo2_proofs 0.2.0 `plonk::Expression<F>` (plonk/circuit.rs) for the quotient numerator that
`create_proof` evaluates over the extended domain (reference call site /root/reference/src/test_utils.rs:41-49; the
reference's gates are built from exactly these nodes in src/circuits/tables/exe.rs:147-498, logic.rs:125-185,
sprod.rs:65-92):

    Constant(v) | Selector(i) | Fixed(col, rot) | Advice(col, rot) | Instance(col, rot)
    Negated(e) | Sum(a, b) | Product(a, b) | Scaled(e, v)        with + - * and unary - overloaded, .degree()

`compile_gates` lowers a list of gate polynomials to libtrh's stack program (include/trh.h, TRH_EXPR_*): each gate is
evaluated depth-first (the operand that needs the deeper stack first, so the stack stays shallow), then folded into
the accumulator with the challenge y -- h = h * y + gate, the order `create_proof` uses -- and `GateEvaluator` runs the
program on device-resident columns.  Values are plain ints here; limbs only appear at the ABI.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import api
from .poly import _MODULUS

OP = dict(PUSH_COLUMN=0, PUSH_CONST=1, PUSH_LOCAL=2, ADD=3, SUB=4, MUL=5, NEG=6, SQR=7, MUL_CONST=8, ADD_CONST=9,
          STORE_LOCAL=10, FOLD=11, STORE_TOP=12, STORE_ACC=13)


class Expression:
    def __add__(self, o):
        return Sum(self, _wrap(o))

    def __radd__(self, o):
        return Sum(_wrap(o), self)

    def __sub__(self, o):
        return Sum(self, Negated(_wrap(o)))  # halo2: a - b == a + (-b)

    def __rsub__(self, o):
        return Sum(_wrap(o), Negated(self))

    def __mul__(self, o):
        return Scaled(self, o) if isinstance(o, int) else Product(self, o)

    def __rmul__(self, o):
        return Scaled(self, o) if isinstance(o, int) else Product(o, self)

    def __neg__(self):
        return Negated(self)

    def degree(self) -> int:
        raise NotImplementedError


def _wrap(o):
    return Constant(o) if isinstance(o, int) else o


@dataclass(frozen=True, eq=False)
class Constant(Expression):
    value: int

    def degree(self):
        return 0


@dataclass(frozen=True, eq=False)
class _Query(Expression):
    column: int
    rotation: int = 0
    kind = "advice"

    def degree(self):
        return 1


class Advice(_Query):
    kind = "advice"


class Fixed(_Query):
    kind = "fixed"


class Instance(_Query):
    kind = "instance"


class Selector(_Query):  # a selector is a fixed column queried at the current row
    kind = "selector"


@dataclass(frozen=True, eq=False)
class Negated(Expression):
    e: Expression

    def degree(self):
        return self.e.degree()


@dataclass(frozen=True, eq=False)
class Sum(Expression):
    a: Expression
    b: Expression

    def degree(self):
        return max(self.a.degree(), self.b.degree())


@dataclass(frozen=True, eq=False)
class Product(Expression):
    a: Expression
    b: Expression

    def degree(self):
        return self.a.degree() + self.b.degree()


@dataclass(frozen=True, eq=False)
class Scaled(Expression):
    e: Expression
    value: int

    def degree(self):
        return self.e.degree()


# ---------------------------------------------------------------------------------------
# lowering
# ---------------------------------------------------------------------------------------
def _need(e) -> int:
    """stack entries an expression needs (Sethi-Ullman number)"""
    if isinstance(e, (Constant, _Query)):
        return 1
    if isinstance(e, (Negated, Scaled)):
        return _need(e.e)
    a, b = _need(e.a), _need(e.b)
    return max(a, b) if a != b else a + 1


@dataclass
class Program:
    field: str
    insns: np.ndarray          # (n, 3) int64: op, a, rotation
    consts: list               # ints (index 0 is the folding challenge y)
    columns: list              # (kind, column) of every resident column slot, in slot order
    max_degree: int


class _Lowering:
    def __init__(self, field: str, first_const: int | None = None):
        self.m = _MODULUS[field]
        self.consts, self.const_ix = ([first_const % self.m] if first_const is not None else []), {}
        self.columns, self.col_ix = [], {}
        self.insns = []

    def const(self, v):
        v %= self.m
        if v not in self.const_ix:
            self.const_ix[v] = len(self.consts)
            self.consts.append(v)
        return self.const_ix[v]

    def column(self, q):
        key = (q.kind, q.column)
        if key not in self.col_ix:
            self.col_ix[key] = len(self.columns)
            self.columns.append(key)
        return self.col_ix[key]

    def emit(self, e):
        insns, emit = self.insns, self.emit
        if isinstance(e, Constant):
            insns.append((OP["PUSH_CONST"], self.const(e.value), 0))
        elif isinstance(e, _Query):
            insns.append((OP["PUSH_COLUMN"], self.column(e), e.rotation))
        elif isinstance(e, Negated):
            emit(e.e)
            insns.append((OP["NEG"], 0, 0))
        elif isinstance(e, Scaled):
            emit(e.e)
            insns.append((OP["MUL_CONST"], self.const(e.value), 0))
        elif isinstance(e, Sum):
            if isinstance(e.b, Negated):  # a + (-b): one SUB instead of NEG + ADD
                a, b = e.a, e.b.e
                if _need(b) > _need(a):   # b first, then a: top = a, next = b -> b - a, negate
                    emit(b); emit(a)
                    insns.append((OP["SUB"], 0, 0)); insns.append((OP["NEG"], 0, 0))
                else:
                    emit(a); emit(b)
                    insns.append((OP["SUB"], 0, 0))
            else:
                first, second = (e.b, e.a) if _need(e.b) > _need(e.a) else (e.a, e.b)
                emit(first); emit(second)
                insns.append((OP["ADD"], 0, 0))
        elif isinstance(e, Product):
            if e.a is e.b:
                emit(e.a)
                insns.append((OP["SQR"], 0, 0))
            else:
                first, second = (e.b, e.a) if _need(e.b) > _need(e.a) else (e.a, e.b)
                emit(first); emit(second)
                insns.append((OP["MUL"], 0, 0))
        else:
            raise TypeError(f"not an Expression: {e!r}")

    def program(self, field, max_degree):
        return Program(field, np.array(self.insns, dtype=np.int64).reshape(-1, 3), self.consts, self.columns, max_degree)


def compile_gates(field: str, gates, y: int = 1) -> Program:
    """gates: iterable of Expression, folded into ONE output with the challenge y (h = h * y + gate).
    Constant 0 of the program is y (GateEvaluator.set_challenge replaces it)."""
    lo = _Lowering(field, first_const=y)
    max_degree = 0
    for g in gates:
        max_degree = max(max_degree, g.degree())
        lo.emit(g)
        lo.insns.append((OP["FOLD"], 0, 0))  # acc = acc * y + gate
    lo.insns.append((OP["STORE_ACC"], 0, 0))
    return lo.program(field, max_degree)


def compile_outputs(field: str, exprs) -> Program:
    """one output column per expression (e.g. the numerator and denominator of a grand-product argument)"""
    lo = _Lowering(field)
    max_degree = 0
    for i, e in enumerate(exprs):
        max_degree = max(max_degree, e.degree())
        lo.emit(e)
        lo.insns.append((OP["STORE_TOP"], i, 0))
    return lo.program(field, max_degree)


def _limbs(field: str, v: int) -> np.ndarray:
    m = _MODULUS[field]
    x = v % m * ((1 << 256) % m) % m
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class _Insn(ctypes.Structure):
    _fields_ = [("op", ctypes.c_uint32), ("a", ctypes.c_uint32), ("rotation", ctypes.c_int32)]


class GateEvaluator:
    """A compiled gate set resident on the device.  eval(columns, log_n, rot_step) -> h numerator values."""

    def __init__(self, program: Program, n_outputs: int = 1, n_locals: int = 0):
        self.program = program
        arr = (_Insn * len(program.insns))()
        for i, (op, a, rot) in enumerate(program.insns):
            arr[i] = _Insn(int(op), int(a), int(rot))
        consts = np.stack([_limbs(program.field, v) for v in program.consts]) if program.consts else np.zeros((1, 4), np.uint64)
        h = ctypes.c_void_p()
        api._check(api.lib().trh_expr_create(api.FIELD_ID[program.field], ctypes.cast(arr, ctypes.c_void_p), len(program.insns), api._p(consts), len(program.consts),
                                            len(program.columns), n_outputs, n_locals, ctypes.byref(h)))
        self.handle, self.n_outputs = h, n_outputs

    def lds_slots(self) -> int:
        return int(api.lib().trh_expr_lds_slots(self.handle))

    def set_challenge(self, y: int):
        """constant 0 of the program: the gate-folding challenge y of this proof"""
        api._check(api.lib().trh_expr_set_const(self.handle, 0, api._p(_limbs(self.program.field, y))))

    def eval(self, columns: dict, log_n: int, rot_step: int = 1, out=None, stream=None):
        """columns: {(kind, column): device tensor (2^log_n, 4)}; returns the output tensor (2^log_n, 4)"""
        import torch
        first = next(iter(columns.values()))
        n = 1 << log_n
        ptrs = (ctypes.c_void_p * max(1, len(self.program.columns)))()
        for i, key in enumerate(self.program.columns):
            t = columns[key]
            assert t.shape[-2] == n and t.is_contiguous(), key
            ptrs[i] = t.data_ptr()
        if out is None:
            out = torch.empty((self.n_outputs, n, 4), dtype=first.dtype, device=first.device)
        o3 = out.reshape(self.n_outputs, n, 4)
        outs = (ctypes.c_void_p * self.n_outputs)(*[o3[i].data_ptr() for i in range(self.n_outputs)])
        if stream is None:
            stream = torch.cuda.current_stream(first.device).cuda_stream
        api._check(api.lib().trh_expr_eval_dev(self.handle, ptrs, outs, log_n, rot_step, stream))
        return o3[0] if self.n_outputs == 1 else o3

    def eval_blocks(self, columns: dict, block_log: int, n_blocks: int, out=None, stream=None):
        """the same over columns in the coset-block layout of EvaluationDomain.coeff_to_extended_blocks: (n_blocks, 2^block_log, 4)
        per column; Rotation(r) reads row q + r of the same block.  Returns (n_blocks, 2^block_log, 4) (a leading output axis when
        the program has several outputs)"""
        import torch
        first = next(iter(columns.values()))
        rows = n_blocks << block_log
        ptrs = (ctypes.c_void_p * max(1, len(self.program.columns)))()
        for i, key in enumerate(self.program.columns):
            t = columns[key]
            assert t.numel() == rows * 4 and t.is_contiguous(), key
            ptrs[i] = t.data_ptr()
        if out is None:
            out = torch.empty((self.n_outputs, n_blocks, 1 << block_log, 4), dtype=first.dtype, device=first.device)
        o4 = out.reshape(self.n_outputs, n_blocks, 1 << block_log, 4)
        outs = (ctypes.c_void_p * self.n_outputs)(*[o4[i].data_ptr() for i in range(self.n_outputs)])
        if stream is None:
            stream = torch.cuda.current_stream(first.device).cuda_stream
        api._check(api.lib().trh_expr_eval_blocks_dev(self.handle, ptrs, outs, block_log, n_blocks, stream))
        return o4[0] if self.n_outputs == 1 else o4

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                api.lib().trh_expr_destroy(self.handle)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------
# synthetic gate set with the shape of TinyRamCircuit's (Appendix B of SURVEY.md): selector-gated constraints of
# degree <= 6 over advice columns at rotations 0 / +1, a few range-style products and linear relations
# ---------------------------------------------------------------------------------------
def synthetic_gates(n_advice: int, n_fixed: int, n_gates: int, seed: int = 0x6A7E):
    import random
    rng = random.Random(seed)

    def adv():
        return Advice(rng.randrange(n_advice), rng.choice((0, 0, 0, 1, -1)))

    gates = []
    for g in range(n_gates):
        sel = Selector(rng.randrange(n_fixed))
        kind = g % 4
        if kind == 0:    # linear relation: s * (a + 2^16 b - c)
            body = adv() + adv() * (1 << 16) - adv()
        elif kind == 1:  # product check: s * (a * b - c)
            body = adv() * adv() - adv()
        elif kind == 2:  # boolean / small-range: s * v (1 - v) (2 - v)
            v = adv()
            body = v * (Constant(1) - v) * (Constant(2) - v)
        else:            # the degree-6 shape of sprod: s^2-like gating of two quadratic factors
            a, b = adv(), adv()
            body = (a * a - adv()) * (b * b - adv()) * (adv() - 1)
        gates.append(sel * body)
    return gates
