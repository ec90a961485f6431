import os, sys, random
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cpu_ref, pasta as o
from tiny_ram_halo2_amd import api, synth, poly, ipa, expr
api.init(0)
bad = 0
def aff(curve, jac): return cpu_ref.to_affine(curve, jac)
# 1. tiny MSMs, identity-only bases, zero scalars, repeated base (doubling / cancellation inside one bucket)
for curve in ("pallas", "vesta"):
    f = o.CURVES[curve].scalar
    for n in (1, 2, 3, 5, 17, 64, 65, 257):
        bases = cpu_ref.gen_bases(curve, 7, 3, n, threads=1)
        for kind in ("rand", "zero", "ident", "same", "neg"):
            b = bases.copy(); sc = synth.field_elements(n * 31 + len(kind), n).copy()
            if kind == "zero": sc[:] = 0
            if kind == "ident": b[:] = 0
            if kind == "same": b[:] = b[0]; sc[:] = np.array(f.limbs(5), np.uint64)          # n copies of the same point in one bucket
            if kind == "neg":
                b[:] = b[0]; sc[0::2] = np.array(f.limbs(9), np.uint64); sc[1::2] = np.array(f.limbs(f.m - 9), np.uint64)  # +9 P, -9 P alternating
            h = api.Bases.from_host(curve, b)
            got = h.msm(sc)
            want = aff(curve, cpu_ref.best_multiexp(curve, sc, b, threads=2))
            if not (got[:8] == want).all(): bad += 1; print("MSM mismatch", curve, n, kind)
            if n >= 17:
                h.precompute(0)
                if not (h.msm(sc)[:8] == want).all(): bad += 1; print("fixed-base MSM mismatch", curve, n, kind)
print("msm edge cases done, bad =", bad)
# 2. expr: locals holding big sums, stored sums, FOLD of big values
f = o.FIELDS["fq"]
I = expr._Insn; OP = expr.OP
import ctypes
n = 8
cols = {("advice", c): [f.m - 1 - r * c for r in range(n)] for c in range(3)}
dev = [torch.from_numpy(np.array([f.limbs(v) for v in cols[("advice", c)]], dtype=np.uint64).view(np.int64)).cuda() for c in range(3)]
ins = []
for rep in range(12):  # T = sum of 12 columns (bound 384 -> reductions), kept in a local
    ins.append((OP["PUSH_COLUMN"], rep % 3, 0))
    if rep: ins.append((OP["ADD"], 0, 0))
ins += [(OP["STORE_LOCAL"], 0, 0), (OP["PUSH_LOCAL"], 0, 0), (OP["MUL"], 0, 0),      # S * S
        (OP["PUSH_LOCAL"], 0, 0), (OP["ADD"], 0, 0), (OP["PUSH_LOCAL"], 0, 0), (OP["ADD"], 0, 0),  # + S + S
        (OP["STORE_TOP"], 0, 0)]
arr = (I * len(ins))(*[I(*t) for t in ins])
h = ctypes.c_void_p()
one = np.zeros((1, 4), np.uint64)
api._check(api.lib().trh_expr_create(api.FIELD_ID["fq"], ctypes.cast(arr, ctypes.c_void_p), len(ins), api._p(one), 1, 3, 1, 1, ctypes.byref(h)))
out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in dev]); outs = (ctypes.c_void_p * 1)(out.data_ptr())
api._check(api.lib().trh_expr_eval_dev(h, ptrs, outs, 3, 1, None))
torch.cuda.synchronize()
got = [f.from_limbs(r) for r in out.cpu().numpy().view(np.uint64)]
for r in range(n):
    S = sum(cols[("advice", rep % 3)][r] for rep in range(12)) % f.m
    if got[r] != (S * S + 2 * S) % f.m: bad += 1; print("expr mismatch row", r)
print("expr edge done, bad =", bad)
# 3. IPA with tables at k = 1, 2, 3
from common import OracleTranscript
for k in (1, 2, 3):
    curve = "vesta"; cv = o.CURVES[curve]; fs = cv.scalar; nn = 1 << k
    g_l = cpu_ref.gen_bases(curve, 17, 5, nn, threads=1); w_l = cpu_ref.gen_bases(curve, 99, 1, 1, threads=1); u_l = cpu_ref.gen_bases(curve, 77, 1, 1, threads=1)
    res = []
    for pre in (False, True):
        params = poly.Params(curve, k, g_l, g_l, w_l, u=u_l, precompute=pre)
        rnd = random.Random(5 + k)
        p_l = synth.field_elements(3 + k, nn); s_l = synth.field_elements(9 + k, nn)
        draws = iter([rnd.randrange(fs.m) for _ in range(2 * k)])
        class T:
            def __init__(s): s.log = []; s.c = 1234567
            def write_point(s, p): s.log.append(("p", np.asarray(p)[:8].tolist()))
            def write_scalar(s, x): s.log.append(("s", np.asarray(x).tolist()))
            def squeeze_challenge_scalar(s): s.c = (s.c * 6364136223846793005 + 1442695040888963407) % fs.m; return s.c
        t = T()
        try:
            c, ff = ipa.create_proof_native(params, lambda: next(draws), t, torch.from_numpy(p_l.view(np.int64)).cuda(), 11, 13, s_l, 17)
            res.append((c, ff, t.log))
        except Exception as e:
            res.append(("err", str(e)))
    if res[0] != res[1]: bad += 1; print("IPA forms differ at k =", k, res[0][:2] if res[0][0] != "err" else res[0], res[1][:2] if res[1][0] != "err" else res[1])
print("ALL DONE bad =", bad)
