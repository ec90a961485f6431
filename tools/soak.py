"""Randomised self-consistency soak of libtrh (no oracle: different code paths of the library must agree bit for bit).
tools/soak.py [seconds] [seed]"""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, permutation, poly, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
api.init(0)
MOD = {"fp": poly._MODULUS["fp"], "fq": poly._MODULUS["fq"]}
t_end = time.time() + budget
stats = {"msm": 0, "ntt": 0, "lookup": 0}
fails = 0


def limbs(field, v):
    m = MOD[field]
    x = v % m * ((1 << 256) % m) % m
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def scalars(field, n, kind):
    a = synth.field_elements(rng.randrange(1 << 30), n).copy()
    m = MOD[field]
    if kind == "small":
        a[:] = np.array([limbs(field, v) for v in (rng.randrange(1 << 16) for _ in range(min(n, 64)))], dtype=np.uint64)[np.arange(n) % min(n, 64)]
    elif kind == "same":
        a[:] = np.array(limbs(field, rng.randrange(m)), dtype=np.uint64)
    elif kind == "edge":
        pool = np.array([limbs(field, v) for v in (0, 1, m - 1, m - 2, 2, (m - 1) // 2)], dtype=np.uint64)
        idx = np.array([rng.randrange(6) for _ in range(n)])
        mask = np.array([rng.random() < 0.5 for _ in range(n)])
        a[mask] = pool[idx[mask]]
    return a


while time.time() < t_end:
    which = rng.choice(["msm", "msm", "ntt", "lookup"])
    if which == "msm":
        curve = rng.choice(["pallas", "vesta"])
        sf = api.SCALAR_FIELD[curve]
        n = rng.choice([rng.randrange(1, 200), rng.randrange(200, 1 << 14), rng.randrange(1 << 14, 1 << 19)])
        xy = api.Bases.generate(curve, rng.randrange(1, 1 << 40), rng.randrange(1, 1 << 30), n).download()
        for _ in range(rng.randrange(0, 4)):  # identity / duplicate / opposite bases
            i, j = rng.randrange(n), rng.randrange(n)
            r = rng.random()
            if r < 0.4:
                xy[i] = 0
            else:
                xy[i] = xy[j]
        bases = api.Bases.from_host(curve, xy)
        sc = scalars(sf, n, rng.choice(["uniform", "uniform", "small", "same", "edge"]))
        d = torch.from_numpy(sc.view(np.int64)).cuda()
        ref = bases.msm_dev(d, n)
        api.set_window_bits(rng.randrange(2, 19))
        forced = bases.msm_dev(d, n)
        api.set_window_bits(0)
        ok = (forced == ref).all()
        if n >= 2:
            cut = rng.randrange(1, n)
            parts = np.stack([bases.msm_dev(d, cut), bases.msm_dev(d[cut:].contiguous(), n - cut, offset=cut)])
            ok = ok and (api.point_sum(curve, parts) == ref).all()
        try:
            bases.precompute(rng.choice([0, 0, rng.randrange(6, 19)]))
            ok = ok and (bases.msm_dev(d, n) == ref).all()
            b2 = bases.msm_batch_dev(torch.stack([d, d]).contiguous(), n, 2)
            ok = ok and (b2[0] == ref).all() and (b2[1] == ref).all()
        except api.TrhError:
            pass
        ok = ok and (api.best_multiexp(curve, sc, xy) == ref).all()
        if not ok:
            fails += 1
            print("MSM MISMATCH", curve, n, flush=True)
    elif which == "ntt":
        field = rng.choice(["fp", "fq"])
        k = rng.randrange(1, 21)
        batch = rng.randrange(1, 4)
        dom = poly.EvaluationDomain(field, rng.choice([2, 3, 4, 6]), k)
        n = 1 << k
        a = synth.field_elements(rng.randrange(1 << 30), batch * n).reshape(batch, n, 4)
        d = torch.from_numpy(a.view(np.int64).copy()).cuda()
        lag = dom.coeff_to_lagrange(d.clone())
        back = dom.lagrange_to_coeff(lag)
        ok = (back.cpu().numpy().view(np.uint64) == a).all()
        if dom.extended_k <= 22:
            ext = dom.coeff_to_extended(d)
            bk = dom.extended_to_coeff(ext).cpu().numpy().view(np.uint64)
            ok = ok and (bk[:, :n] == a).all() and (bk[:, n:] == 0).all()
        if not ok:
            fails += 1
            print("NTT MISMATCH", field, k, batch, flush=True)
    else:
        field = rng.choice(["fp", "fq"])
        n = rng.choice([rng.randrange(1, 100), rng.randrange(100, 1 << 12), rng.randrange(1 << 12, 1 << 17)])
        tsize = rng.randrange(1, n + 1)
        wide = rng.random() < 0.5
        m = MOD[field]
        vals = np.array([limbs(field, rng.randrange(m) if wide else rng.randrange(1 << 20)) for _ in range(min(tsize, 512))], dtype=np.uint64)
        table = vals[np.arange(n) % len(vals)]
        inp = vals[np.array([rng.randrange(len(vals)) for _ in range(n)])]
        dt, di = torch.from_numpy(table.view(np.int64)).cuda(), torch.from_numpy(inp.view(np.int64)).cuda()
        pa, ps = permutation.lookup_permute(field, di, dt)
        pa_h, ps_h = pa.cpu().numpy().view(np.uint64), ps.cpu().numpy().view(np.uint64)
        eq_prev = np.concatenate([[False], (pa_h[1:] == pa_h[:-1]).all(axis=1)])
        ok = ((pa_h == ps_h).all(axis=1) | eq_prev).all()
        ok = ok and sorted(map(bytes, pa_h)) == sorted(map(bytes, inp)) and sorted(map(bytes, ps_h)) == sorted(map(bytes, table))
        if not ok:
            fails += 1
            print("LOOKUP MISMATCH", field, n, tsize, flush=True)
    stats[which] += 1
print("soak:", stats, "failures:", fails)
sys.exit(1 if fails else 0)
