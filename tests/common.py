"""Shared helpers for the test-suite: golden-vector loading and limb conversions."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def unhex(limbs):
    return np.array([int(v, 16) for v in limbs], dtype=np.uint64)


def unhex_rows(rows):
    return np.array([[int(v, 16) for v in r] for r in rows], dtype=np.uint64)


# ---- transcripts of the IPA / multiopen parity tests: a stand-in for Blake2bWrite whose challenges hash everything written so far;
#      the device side and the C++ oracle write limb arrays, the big-int oracle writes ints / affine tuples -- same bytes -------------
class HashTranscript:
    """stand-in for Blake2bWrite: challenges are a hash of everything written so far"""

    def __init__(self, modulus):
        self.h, self.m, self.log = hashlib.blake2b(b"trh-test-transcript"), modulus, []

    def _absorb(self, tag, data):
        self.h.update(tag + bytes(data))
        self.log.append((tag, bytes(data)))

    def squeeze_challenge_scalar(self):
        self.h.update(b"challenge")
        return int.from_bytes(self.h.digest(), "little") % self.m


class DeviceTranscript(HashTranscript):
    def write_point(self, jac):
        self._absorb(b"P", np.ascontiguousarray(jac, dtype=np.uint64)[:8].tobytes())

    def write_scalar(self, limbs):
        self._absorb(b"S", np.ascontiguousarray(limbs, dtype=np.uint64).tobytes())


class OracleTranscript(HashTranscript):
    def __init__(self, curve):
        super().__init__(curve.scalar.m)
        self.curve = curve

    def write_point(self, pt):
        self._absorb(b"P", np.array(self.curve.affine_limbs(pt), dtype=np.uint64).tobytes())

    def write_scalar(self, v):
        self._absorb(b"S", np.array(self.curve.scalar.limbs(v), dtype=np.uint64).tobytes())


class LimbTranscript(DeviceTranscript):
    """DeviceTranscript whose challenges are returned as Montgomery limb arrays (what the C++ oracle's callbacks want)"""

    def __init__(self, field):
        super().__init__(field.m)
        self.field = field

    def squeeze_challenge_scalar(self):
        return np.array(self.field.limbs(super().squeeze_challenge_scalar()), dtype=np.uint64)


# ---- fast (C++ oracle) forms of the verifier equation and of multiopen, for k = 10 .. 18 ---------------------------------------
def ipa_verify_fast(curve, k, g_l, w_l, u_l, commitment_xy, x3, v, s_commitment_xy, xi, z, rounds_xy, challenges, c, f):
    """halo2_proofs 0.2.0 poly/commitment/verifier.rs `verify_proof` (the equation oracle/pasta.py::ipa_verify_proof restates in
    big ints), evaluated with the C++ oracle's group arithmetic so that k = 18 takes seconds:
        P - [v] G_0 + [xi] S + sum_j [u_j^-1] L_j + sum_j [u_j] R_j  ==  [c] G'_0 + [c b z] U + [f] W
    Scalars are canonical ints, points 8-limb affine PODs."""
    import cpu_ref
    import pasta as o
    cv = o.CURVES[curve]
    fs, sf = cv.scalar, {"pallas": "fq", "vesta": "fp"}[curve]
    m, n = fs.m, 1 << k
    lim = lambda val: np.array(fs.limbs(val % m), dtype=np.uint64)  # noqa: E731
    # s_i = prod of u_j over the rounds j whose fold put index i in the upper half (compute_s)
    s = np.tile(lim(1), (n, 1))
    idx = np.arange(n)
    for j, u_j in enumerate(challenges):
        sel = ((idx >> (k - 1 - j)) & 1) == 1
        s[sel] = cpu_ref.field_op(sf, "mul", s[sel], np.tile(lim(u_j), (int(sel.sum()), 1)))
    # b = sum_i s_i x3^i (compute_b): powers by doubling, then a tree of additions
    pw = np.tile(lim(1), (n, 1))
    step, span = x3 % m, 1
    while span < n:  # pw[i + span] = pw[i] * x3^span
        pw[span:2 * span] = cpu_ref.field_op(sf, "mul", pw[:span], np.tile(lim(step), (span, 1)))
        step, span = step * step % m, span * 2
    terms = cpu_ref.field_op(sf, "mul", s, pw)
    while terms.shape[0] > 1:
        h = terms.shape[0] // 2
        terms = cpu_ref.field_op(sf, "add", terms[:h], terms[h:])
    b = fs.from_limbs(terms[0])
    g0 = cpu_ref.best_multiexp(curve, s, g_l, threads=cpu_ref.hardware_threads())

    def jac(xy):
        out = np.zeros(12, dtype=np.uint64)
        xy = np.asarray(xy, dtype=np.uint64).reshape(8)
        if xy.any():
            out[:8] = xy
            out[8:] = np.array(cv.base.limbs(1), dtype=np.uint64)
        return out

    def combo(pairs):  # sum of scalar * point
        pts = np.stack([np.asarray(p, dtype=np.uint64).reshape(8) for _, p in pairs])
        scaled = cpu_ref.scale_points_each(curve, pts, np.stack([lim(sv) for sv, _ in pairs]), threads=4)
        acc = np.zeros(12, dtype=np.uint64)
        for r in scaled:
            acc = cpu_ref.point_op(curve, "add", acc, jac(r))
        return acc

    lhs_pairs = [(1, commitment_xy), (-v, g_l[0]), (xi, s_commitment_xy)]
    for (l_xy, r_xy), u_j in zip(rounds_xy, challenges):
        lhs_pairs += [(pow(u_j, -1, m), l_xy), (u_j, r_xy)]
    lhs = cpu_ref.to_affine(curve, combo(lhs_pairs))
    rhs = cpu_ref.to_affine(curve, combo([(c, cpu_ref.to_affine(curve, g0)), (c * b % m * z % m, u_l), (f, w_l)]))
    return bool((lhs == rhs).all())


def multiopen_create_proof_fast(curve, k, g_l, w_l, u_l, rng, transcript, queries, polys, blinds):
    """oracle/pasta.py::multiopen_create_proof (poly::multiopen::create_proof) with the same point-set construction and integer
    polynomial arithmetic, the commitment and the IPA through the C++ oracle: usable at k >= 10.  `transcript` is a LimbTranscript
    (limb arrays in, limb challenges out), rng() -> int.  polys[key]: list of ints."""
    import cpu_ref
    import pasta as o
    cv = o.CURVES[curve]
    f_, sf = cv.scalar, {"pallas": "fq", "vesta": "fp"}[curve]
    m, n = f_.m, 1 << k
    lim = lambda val: np.array(f_.limbs(val % m), dtype=np.uint64)  # noqa: E731
    arr = lambda poly: np.array([f_.limbs(v) for v in poly], dtype=np.uint64)  # noqa: E731
    sq = lambda: f_.from_limbs(transcript.squeeze_challenge_scalar())  # noqa: E731
    x1, x2 = sq(), sq()
    point_index, commitment_points, order = {}, {}, []
    for point, key in queries:
        idx = point_index.setdefault(point, len(point_index))
        if key not in commitment_points:
            commitment_points[key] = []
            order.append(key)
        commitment_points[key].append(idx)
    inverse = {i: p for p, i in point_index.items()}
    set_index, set_of = {}, {}
    for key in order:
        s_ = tuple(sorted(set(commitment_points[key])))
        set_of[key] = set_index.setdefault(s_, len(set_index))
    point_sets = [None] * len(set_index)
    for s_, i in set_index.items():
        point_sets[i] = [inverse[j] for j in s_]
    q_polys, q_blinds = [None] * len(point_sets), [0] * len(point_sets)
    for key in order:
        i = set_of[key]
        q_polys[i] = list(polys[key]) if q_polys[i] is None else [(a * x1 + b) % m for a, b in zip(q_polys[i], polys[key])]
        q_blinds[i] = (q_blinds[i] * x1 + blinds[key]) % m
    q_prime = None
    for pts, q in zip(point_sets, q_polys):
        cur = q
        for zz in pts:
            cur = o.kate_division(f_, cur, zz)
        cur = cur + [0] * (n - len(cur))
        q_prime = cur if q_prime is None else [(a * x2 + b) % m for a, b in zip(q_prime, cur)]
    q_prime_blind = rng()
    bases = np.concatenate([np.asarray(g_l, dtype=np.uint64).reshape(n, 8), np.asarray(w_l, dtype=np.uint64).reshape(1, 8)])
    com = cpu_ref.best_multiexp(curve, np.concatenate([arr(q_prime), lim(q_prime_blind)[None]]), bases, threads=cpu_ref.hardware_threads())
    com_xy = cpu_ref.to_affine(curve, com)
    norm = np.zeros(12, dtype=np.uint64)
    if com_xy.any():
        norm[:8] = com_xy
        norm[8:] = np.array(cv.base.limbs(1), dtype=np.uint64)
    transcript.write_point(norm)
    x3 = sq()
    for q in q_polys:
        transcript.write_scalar(cpu_ref.eval_polynomial(sf, arr(q), lim(x3)))
    x4 = sq()
    p_poly, p_blind = q_prime, q_prime_blind
    for q, b in zip(q_polys, q_blinds):
        p_poly = [(a * x4 + c) % m for a, c in zip(p_poly, q)]
        p_blind = (p_blind * x4 + b) % m
    s_poly = [rng() for _ in range(n)]
    s_blind = rng()  # drawn before the opening's own randomness, as the big-int restatement does
    c, f = cpu_ref.ipa_create_proof(curve, k, g_l, w_l, u_l, lambda: lim(rng()), transcript, arr(p_poly), lim(p_blind), lim(x3), arr(s_poly), lim(s_blind))
    return f_.from_limbs(c), f_.from_limbs(f)


# ---- gate fixtures (tests/golden/*_gates.json): JSON expression trees <-> the Python Expression mirror <-> the oracle's tuple form ----
def expr_from_json(j):
    from tiny_ram_halo2_amd import expr
    tag = j[0]
    if tag == "const":
        return expr.Constant(int(j[1], 16))
    if tag in ("advice", "fixed", "instance", "selector"):
        cls = {"advice": expr.Advice, "fixed": expr.Fixed, "instance": expr.Instance, "selector": expr.Selector}[tag]
        return cls(j[1], j[2])
    if tag == "neg":
        return expr.Negated(expr_from_json(j[1]))
    if tag == "sum":
        return expr.Sum(expr_from_json(j[1]), expr_from_json(j[2]))
    if tag == "prod":
        return expr.Product(expr_from_json(j[1]), expr_from_json(j[2]))
    if tag == "scaled":
        return expr.Scaled(expr_from_json(j[1]), int(j[2], 16))
    raise ValueError(tag)


def expr_to_tuple(e):
    from tiny_ram_halo2_amd import expr
    if isinstance(e, expr.Constant):
        return ("const", e.value)
    if isinstance(e, expr._Query):
        return ("col", (e.kind, e.column), e.rotation)
    if isinstance(e, expr.Negated):
        return ("neg", expr_to_tuple(e.e))
    if isinstance(e, expr.Sum):
        return ("sum", expr_to_tuple(e.a), expr_to_tuple(e.b))
    if isinstance(e, expr.Product):
        return ("prod", expr_to_tuple(e.a), expr_to_tuple(e.b))
    return ("scaled", expr_to_tuple(e.e), e.value)


# ---- library options are fixed while a context exists (trh_set_option / TRH_<NAME> read once): a test that needs another setting runs the
#      computation in a fresh process and compares what it prints with this process's result -----------------------------------------------
def run_with_options(script: str, env: dict, timeout: int = 600) -> str:
    """runs `script` (python source; the repo root and tests/ are on sys.path) in a child process with `env` added; returns its stdout"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prologue = ("import sys\n" + "".join(f"sys.path.insert(0, {p!r})\n" for p in (os.path.join(root, "tests"), os.path.join(root, "oracle"), root)))
    r = subprocess.run([sys.executable, "-c", prologue + script], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env), cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def point_hex(p) -> str:
    return "".join(f"{int(v):016x}" for v in np.asarray(p, dtype=np.uint64).reshape(-1))
