"""The grand-product column of halo2_proofs 0.2.0's permutation argument (plonk/permutation/prover.rs `Argument::commit`,
reached from create_proof: /root/reference/src/test_utils.rs:41-49; the reference enables equality on 188 columns,
src/circuits/tables/prog.rs:151-152, i.e. 47 product columns of 4 columns each -- SURVEY.md Appendix B) on device columns:

    z[0] = z0,   z[i + 1] = z[i] * prod_j (v_j[i] + beta * delta^(j0 + j) * omega^i + gamma) / (v_j[i] + beta * sigma_j[i] + gamma)

built from libtrh primitives: the numerator and denominator products are one two-output expression program
(expr.compile_outputs), the division is ff::BatchInvert (trh_field_batch_invert_dev) + an element-wise multiply, the
running product is trh_field_prefix_product_dev.  Blinding rows and the commitment of z are the caller's (the replay
commits product columns as ordinary Lagrange columns)."""
from __future__ import annotations

import numpy as np

from . import api, expr
from .poly import _MODULUS, _ROOT_OF_UNITY, S


def delta(field: str) -> int:
    """pasta_curves `DELTA` = GENERATOR^(2^S), GENERATOR = 5: generator of the odd-order part of the multiplicative group"""
    return pow(5, 1 << S, _MODULUS[field])


def omega(field: str, k: int) -> int:
    w = _ROOT_OF_UNITY[field]
    for _ in range(k, S):
        w = w * w % _MODULUS[field]
    return w


class GrandProduct:
    """z[0] = z0, z[i + 1] = z[i] * num(row i) / den(row i) for two Expressions over device columns -- the common core of the
    permutation product above and of the lookup argument's product (plonk/lookup/prover.rs `commit_product`:
    num = (a + beta)(s + gamma) over the compressed input / table, den = (a' + beta)(s' + gamma) over the permuted ones)."""

    def __init__(self, field: str, k: int, num: expr.Expression, den: expr.Expression):
        self.field, self.k, self.n = field, k, 1 << k
        self.ev = expr.GateEvaluator(expr.compile_outputs(field, [num, den]), n_outputs=2)

    def compute(self, columns: dict, z0: int = 1, rot_step: int = 1):
        import torch
        first = next(iter(columns.values()))
        st = torch.cuda.current_stream(first.device).cuda_stream
        nd = self.ev.eval(columns, self.k, rot_step, stream=st)
        num, den = nd[0], nd[1]
        api.batch_invert_dev(self.field, den, self.n, stream=st)
        ratio = torch.empty_like(num)
        api._check(api.lib().trh_field_op_dev(api.FIELD_ID[self.field], api.FIELD_OPS["mul"], api._devptr(num), api._devptr(den), api._devptr(ratio), self.n, st))
        z = torch.empty_like(num)
        api.prefix_product_dev(self.field, ratio, z, self.n, stream=st)
        if z0 % _MODULUS[self.field] != 1:
            api.field_scale_dev(self.field, z, self.n, expr._limbs(self.field, z0), stream=st)
        return z


def grand_products_batch(field: str, k: int, evaluators, column_sets, rot_step: int = 1):
    """All product columns of a proof at once: evaluators[i] (a two-output GateEvaluator: numerator, denominator) over
    column_sets[i]; ONE batch inversion over every denominator (a 255-step exponentiation per 64 elements is latency-bound
    for a single column and free across 78), one multiply, one batched prefix product.  Returns z as (len, n, 4)."""
    import torch
    n, rows = 1 << k, len(evaluators)
    first = next(iter(column_sets[0].values()))
    st = torch.cuda.current_stream(first.device).cuda_stream
    num = torch.empty((rows, n, 4), dtype=first.dtype, device=first.device)
    den = torch.empty_like(num)
    for i, (ev, cols) in enumerate(zip(evaluators, column_sets)):
        nd = ev.eval(cols, k, rot_step, stream=st)
        num[i].copy_(nd[0]); den[i].copy_(nd[1])
    api.batch_invert_dev(field, den, rows * n, stream=st)
    api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS["mul"], api._devptr(num), api._devptr(den), api._devptr(den), rows * n, st))
    z = torch.empty_like(num)
    api._check(api.lib().trh_field_prefix_product_rows_dev(api.FIELD_ID[field], api._devptr(den), api._devptr(z), n, rows, st))
    return z


def permutation_terms(field: str, values, sigmas, omega_powers, beta: int, gamma: int, first_column: int = 0):
    """-> (numerator row, denominator row) of one chunk for grand_products_terms: prod_j (v_j + beta delta^(first_column + j) omega^i + gamma)
    and prod_j (v_j + beta sigma_j + gamma) (plonk/permutation/prover.rs Argument::commit)"""
    m = _MODULUS[field]
    g, b = expr._limbs(field, gamma), expr._limbs(field, beta)
    d = delta(field)
    num = [(v, omega_powers, expr._limbs(field, beta * pow(d, first_column + j, m) % m), g) for j, v in enumerate(values)]
    den = [(v, sg, b, g) for v, sg in zip(values, sigmas)]
    return num, den


def lookup_terms(field: str, a, s_, a_perm, s_perm, beta: int, gamma: int):
    """-> (numerator row, denominator row) of one lookup: (A + beta)(S + gamma) over the compressed input / table, (A' + beta)(S' + gamma)
    over the permuted ones (plonk/lookup/prover.rs commit_product)"""
    b, g = expr._limbs(field, beta), expr._limbs(field, gamma)
    return [(a, None, None, b), (s_, None, None, g)], [(a_perm, None, None, b), (s_perm, None, None, g)]


def grand_products_terms(field: str, k: int, num_rows, den_rows):
    """All product columns of a proof from their term rows (permutation_terms / lookup_terms): ONE launch for every numerator and
    denominator product, ONE batch inversion that also multiplies, one batched prefix product.  Returns z as (len, n, 4)."""
    import torch
    n, rows = 1 << k, len(num_rows)
    assert rows == len(den_rows) and rows > 0
    first = num_rows[0][0][0]
    st = torch.cuda.current_stream(first.device).cuda_stream
    nd = torch.empty((2 * rows, n, 4), dtype=first.dtype, device=first.device)
    api.product_terms_dev(field, list(num_rows) + list(den_rows), n, nd, stream=st)
    num, den = nd[:rows], nd[rows:]
    api.batch_invert_mul_dev(field, den, num, rows * n, stream=st)
    api._check(api.lib().trh_field_prefix_product_rows_dev(api.FIELD_ID[field], api._devptr(den), api._devptr(num), n, rows, st))
    return num


def lookup_product(field: str, k: int, beta: int, gamma: int) -> GrandProduct:
    """columns: ("advice", 0) = compressed input A, 1 = compressed table S, 2 = permuted input A', 3 = permuted table S'"""
    a, s_, ap, sp = (expr.Advice(i, 0) for i in range(4))
    return GrandProduct(field, k, (a + beta) * (s_ + gamma), (ap + beta) * (sp + gamma))


class ProductColumn:
    """compiled once per (field, k, number of columns in the chunk, index of the chunk's first column)"""

    def __init__(self, field: str, k: int, n_columns: int, first_column: int = 0):
        self.field, self.k, self.n, self.n_columns = field, k, 1 << k, n_columns
        m = _MODULUS[field]
        self.d = [pow(delta(field), first_column + j, m) for j in range(n_columns)]
        self._ev = None
        self._key = None

    def _evaluator(self, beta: int, gamma: int):
        if self._key != (beta, gamma):
            m = _MODULUS[self.field]
            x = expr.Fixed(0, 0)  # the column of omega^i
            num = den = None
            for j in range(self.n_columns):
                v, sg = expr.Advice(j, 0), expr.Advice(self.n_columns + j, 0)
                tn = v + x * (beta * self.d[j] % m) + gamma
                td = v + sg * beta + gamma
                num = tn if num is None else num * tn
                den = td if den is None else den * td
            self._ev = expr.GateEvaluator(expr.compile_outputs(self.field, [num, den]), n_outputs=2)
            self._key = (beta, gamma)
        return self._ev

    def columns(self, values, sigmas, omega_powers=None):
        """the column dictionary of this chunk's expression program; omega_powers: the shared column of omega^i"""
        import torch
        assert len(values) == self.n_columns and len(sigmas) == self.n_columns
        dev = values[0].device
        if omega_powers is None:
            omega_powers = torch.empty((self.n, 4), dtype=torch.int64, device=dev)
            api.powers_dev(self.field, omega_powers, self.n, expr._limbs(self.field, omega(self.field, self.k)), stream=torch.cuda.current_stream(dev).cuda_stream)
        cols = {("fixed", 0): omega_powers}
        for j in range(self.n_columns):
            cols[("advice", j)] = values[j]
            cols[("advice", self.n_columns + j)] = sigmas[j]
        return cols

    def evaluator(self, beta: int, gamma: int):
        return self._evaluator(beta, gamma)

    def compute(self, values, sigmas, beta: int, gamma: int, z0: int = 1):
        """values, sigmas: lists of n_columns device tensors (n, 4); returns z as a device tensor (n, 4)"""
        import torch
        dev = values[0].device
        st = torch.cuda.current_stream(dev).cuda_stream
        cols = self.columns(values, sigmas)
        nd = self._evaluator(beta, gamma).eval(cols, self.k, 1, stream=st)
        num, den = nd[0], nd[1]
        api.batch_invert_dev(self.field, den, self.n, stream=st)
        ratio = torch.empty_like(num)
        api._check(api.lib().trh_field_op_dev(api.FIELD_ID[self.field], api.FIELD_OPS["mul"], api._devptr(num), api._devptr(den), api._devptr(ratio), self.n, st))
        z = torch.empty_like(num)
        api.prefix_product_dev(self.field, ratio, z, self.n, stream=st)
        if z0 % _MODULUS[self.field] != 1:
            api.field_scale_dev(self.field, z, self.n, expr._limbs(self.field, z0), stream=st)
        return z


def lookup_permute(field: str, input_col, table_col, usable_rows: int | None = None):
    """plonk/lookup/prover.rs permute_expression_pair on device columns (n, 4): returns (permuted_input, permuted_table) of
    usable_rows rows each; raises api.TrhError when an input value is missing from the table"""
    import torch
    n = input_col.shape[0] if usable_rows is None else usable_rows
    assert table_col.shape[0] >= n and input_col.shape[0] >= n and input_col.is_contiguous() and table_col.is_contiguous()
    a = torch.empty((n, 4), dtype=input_col.dtype, device=input_col.device)
    s = torch.empty_like(a)
    api._check(api.lib().trh_lookup_permute_dev(api.FIELD_ID[field], api._devptr(input_col), api._devptr(table_col), n, api._devptr(a), api._devptr(s),
                                               torch.cuda.current_stream(input_col.device).cuda_stream))
    return a, s


def lookup_permute_batch(field: str, inputs, tables, usable_rows: int | None = None):
    """permute_expression_pair for every lookup of a proof at once: inputs / tables are device tensors (batch, rows, 4) (contiguous);
    returns (permuted_inputs, permuted_tables) of the same shape, rows behind usable_rows zero.  One set of launches and one host
    synchronisation for the batch; raises api.TrhError naming the first lookup with an input value missing from its table"""
    import torch
    batch, rows = inputs.shape[0], inputs.shape[1]
    n = rows if usable_rows is None else usable_rows
    assert tables.shape == inputs.shape and n <= rows and inputs.is_contiguous() and tables.is_contiguous()
    a = torch.zeros_like(inputs)
    s = torch.zeros_like(inputs)
    api._check(api.lib().trh_lookup_permute_batch_dev(api.FIELD_ID[field], api._devptr(inputs), api._devptr(tables), n, rows, batch, api._devptr(a), api._devptr(s),
                                                     torch.cuda.current_stream(inputs.device).cuda_stream))
    return a, s
