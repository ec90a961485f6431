#!/usr/bin/env python3
"""Generates the committed golden vectors from oracle/pasta.py (Python big-int, pinned to the
published pasta_curves constants).  The reference (/root/reference) holds no MSM / NTT / field
known-answer vectors, so these are computed from the mathematical
definitions: field ops mod m, affine group law, MSM = naive sum of double-and-add scalar
multiples, DFT = O(n^2) definition for log_n <= 4 and the radix-2 restatement for log_n = 10.

Run:  python tests/golden/make_golden.py        (rewrites tests/golden/*.json, *.npz)
All limbs are u64 little-endian Montgomery form unless the key says `canonical`.
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pasta as o  # noqa: E402

o.check_published_constants()
rng = random.Random(0x74726821)


def hexl(limbs):
    return ["%016x" % int(v) for v in limbs]


def field_kats():
    out = {}
    for name, f in o.FIELDS.items():
        edge = [0, 1, 2, f.m - 1, f.m - 2, f.R, f.R2, f.m >> 1, (1 << 254) % f.m, f.ROOT_OF_UNITY, f.ZETA]
        vals = edge + [rng.randrange(f.m) for _ in range(21)]
        rows = []
        for i, a in enumerate(vals):
            b = vals[(i * 7 + 3) % len(vals)]
            rows.append(dict(
                a=hexl(f.limbs(a)), b=hexl(f.limbs(b)),
                add=hexl(f.limbs(f.add(a, b))), sub=hexl(f.limbs(f.sub(a, b))),
                mul=hexl(f.limbs(f.mul(a, b))), sqr=hexl(f.limbs(f.sqr(a))),
                neg=hexl(f.limbs(f.neg(a))), inv=hexl(f.limbs(f.inv(a))),
                a_canonical=hexl(o.int_to_limbs(a)),  # == to_repr() as limbs
            ))
        out[name] = dict(modulus=hexl(o.int_to_limbs(f.m)), rows=rows,
                         root_of_unity=hexl(f.limbs(f.ROOT_OF_UNITY)), zeta=hexl(f.limbs(f.ZETA)),
                         two_inv=hexl(f.limbs(f.TWO_INV)), delta=hexl(f.limbs(f.DELTA)))
    return out


def jac_limbs(curve, pt, z=1):
    """affine -> a Jacobian representative with the given Z (X = x z^2, Y = y z^3)."""
    f = curve.base
    if pt is None:
        return f.limbs(0) + f.limbs(0) + f.limbs(0)
    x, y = pt
    return f.limbs(x * z * z) + f.limbs(y * z * z * z) + f.limbs(z)


def curve_kats():
    out = {}
    for name, c in o.CURVES.items():
        G = c.generator
        pts = [c.mul(k, G) for k in (1, 2, 3, 0xDEADBEEF, c.scalar.m - 1, rng.randrange(c.scalar.m), rng.randrange(c.scalar.m))]
        cases = []
        pairs = [(pts[0], pts[1]), (pts[3], pts[5]), (pts[5], pts[5]), (pts[5], c.neg(pts[5])),
                 (None, pts[6]), (pts[6], None), (None, None), (pts[4], pts[0]), (pts[2], pts[6])]
        for p, q in pairs:
            z1, z2 = rng.randrange(1, c.base.m), rng.randrange(1, c.base.m)
            cases.append(dict(
                p_affine=hexl(c.affine_limbs(p)), q_affine=hexl(c.affine_limbs(q)),
                p_jac=hexl(jac_limbs(c, p, z1)), q_jac=hexl(jac_limbs(c, q, z2)),
                sum_affine=hexl(c.affine_limbs(c.add(p, q))),
                dbl_p_affine=hexl(c.affine_limbs(c.double(p))),
            ))
        muls = []
        for k in (0, 1, 2, 5, c.scalar.m - 1, rng.randrange(c.scalar.m)):
            muls.append(dict(k_canonical=hexl(o.int_to_limbs(k)), base=hexl(c.affine_limbs(pts[3])),
                             result_affine=hexl(c.affine_limbs(c.mul(k, pts[3])))))
        out[name] = dict(generator=hexl(c.affine_limbs(G)), add_cases=cases, mul_cases=muls)
    return out


def msm_kats():
    """Full inputs for n <= 33; recipe (synthetic stream + s0/d progression) for n >= 100."""
    out = {}
    for name, c in o.CURVES.items():
        fs = c.scalar
        cases = []
        for n in (1, 2, 3, 4, 31, 32, 33):
            sc = [rng.randrange(fs.m) for _ in range(n)]
            bs = [c.mul(rng.randrange(1, 1 << 40), c.generator) for _ in range(n)]
            if n >= 3:
                sc[0] = 0                      # zero scalar
                sc[1] = fs.m - 1               # max scalar
            if n >= 4:
                bs[2] = None                   # identity base
            if n >= 31:
                bs[5] = bs[4]                  # duplicate base
                bs[7] = c.neg(bs[6]); sc[7] = sc[6]   # P and -P with equal scalars (cancels)
                sc[9] = 1; sc[10] = 1; bs[10] = bs[9]  # forces P + P in one bucket
            res = c.msm_naive(sc, bs)
            assert o.best_multiexp(c, sc, bs) == res
            cases.append(dict(n=n, scalars=[hexl(fs.limbs(s)) for s in sc],
                              bases=[hexl(c.affine_limbs(b)) for b in bs],
                              result_affine=hexl(c.affine_limbs(res))))
        recipes = []
        for n, seed in ((100, 0xA100), (1000, 0xA3E8), (1025, 0xA401)):
            s0, d = 0x1234567, 0x89ABCDEF
            mont = [o.synth_scalar_limbs(seed, i) for i in range(n)]
            sc = [fs.from_limbs(l) for l in mont]   # the stream is the Montgomery memory image
            total = sum(s * (s0 + i * d) for i, s in enumerate(sc)) % fs.m
            res = c.mul(total, c.generator)
            if n == 100:  # cross-check the closed form against the naive definition once
                assert res == c.msm_naive(sc, [c.mul(s0 + i * d, c.generator) for i in range(n)])
            recipes.append(dict(n=n, seed=seed, s0=s0, d=d, result_affine=hexl(c.affine_limbs(res))))
        out[name] = dict(cases=cases, recipes=recipes)
    return out


def ntt_kats():
    arrays, meta = {}, {}
    for name, f in o.FIELDS.items():
        for log_n in (0, 1, 2, 3, 4, 10):
            n = 1 << log_n
            a = [rng.randrange(f.m) for _ in range(n)]
            if log_n >= 2:
                a[0], a[1] = 0, f.m - 1
            w = f.omega(log_n)
            fwd = o.dft_naive(f, a, w) if log_n <= 4 else o.best_fft(f, a, w, log_n)
            if log_n <= 4:
                assert fwd == o.best_fft(f, a, w, log_n)
            winv = f.inv(w)
            inv_unscaled = o.best_fft(f, fwd, winv, log_n)
            ninv = f.inv(n)
            assert [v * ninv % f.m for v in inv_unscaled] == a
            key = f"{name}_{log_n}"
            arrays[key + "_in"] = np.array([f.limbs(v) for v in a], dtype=np.uint64)
            arrays[key + "_fwd"] = np.array([f.limbs(v) for v in fwd], dtype=np.uint64)
            arrays[key + "_inv_unscaled"] = np.array([f.limbs(v) for v in inv_unscaled], dtype=np.uint64)
            meta[key] = dict(omega=hexl(f.limbs(w)), omega_inv=hexl(f.limbs(winv)), n_inv=hexl(f.limbs(ninv)))
    return arrays, meta


def main():
    with open(os.path.join(HERE, "field_kat.json"), "w") as fh:
        json.dump(field_kats(), fh, indent=0)
    with open(os.path.join(HERE, "curve_kat.json"), "w") as fh:
        json.dump(curve_kats(), fh, indent=0)
    with open(os.path.join(HERE, "msm_kat.json"), "w") as fh:
        json.dump(msm_kats(), fh, indent=0)
    arrays, meta = ntt_kats()
    np.savez_compressed(os.path.join(HERE, "ntt_kat.npz"), **arrays)
    with open(os.path.join(HERE, "ntt_kat.json"), "w") as fh:
        json.dump(meta, fh, indent=0)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
