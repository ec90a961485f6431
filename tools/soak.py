"""Randomised soak of libtrh: different code paths of the library must agree bit for bit (no oracle needed), and -- when the caller
passes the oracle module (tests/test_gpu_soak.py does; the product side never imports it) -- a share of the MSM / NTT trials is also
compared with cpu_ref.best_multiexp / best_fft.
    tools/soak.py [seconds] [seed]            stand-alone, self-consistency only
    run(budget_s, seed, oracle=None) -> (stats, failures)   from a test"""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, permutation, poly, synth
MOD = {"fp": poly._MODULUS["fp"], "fq": poly._MODULUS["fq"]}
KINDS = ["msm", "msm", "ntt", "lookup", "blocks", "hostio", "products", "sparse", "padded", "sharded", "opening"]
GROUP = 8  # the suite's device group is [0] * 8 (trh_init_multi accepts only the list it was first given)
rng = random.Random(1)  # re-seeded by run()
cpu_ref = None          # the oracle module when a test hands it over


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


def trial(which, fails):
    """one random trial of kind `which`; a mismatch is printed with its parameters and appended to `fails`"""
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
        if cpu_ref is not None and n <= (1 << 16):  # the oracle's best_multiexp on the same host arrays (incl. the identity / duplicate bases)
            want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc, xy, threads=cpu_ref.hardware_threads()))
            ok = ok and (np.asarray(ref)[:8] == want).all()
            _ORACLE_COUNT[0] += 1
        if not ok:
            fails.append(which)
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
        if cpu_ref is not None and k <= 18:  # coeff_to_lagrange is best_fft with omega: the oracle's transform of the first column
            want = cpu_ref.best_fft(field, a[0], np.asarray(dom._w["omega"], dtype=np.uint64), k, threads=cpu_ref.hardware_threads())
            okf = (lag[0].cpu().numpy().view(np.uint64) == want).all()
            _ORACLE_COUNT[0] += 1
        else:
            okf = True
        back = dom.lagrange_to_coeff(lag)
        ok = okf and (back.cpu().numpy().view(np.uint64) == a).all()
        if dom.extended_k <= 22:
            ext = dom.coeff_to_extended(d)
            bk = dom.extended_to_coeff(ext).cpu().numpy().view(np.uint64)
            ok = ok and (bk[:, :n] == a).all() and (bk[:, n:] == 0).all()
        if not ok:
            fails.append(which)
            print("NTT MISMATCH", field, k, batch, flush=True)
    elif which == "blocks":
        # the coset-block form of the extended domain against the full one (round 3): every block entry, and the quotient back from j - 1 blocks
        field = rng.choice(["fp", "fq"])
        k = rng.randrange(1, 17)
        j = rng.choice([3, 4, 5, 6, 8])
        batch = rng.randrange(1, 4)
        dom = poly.EvaluationDomain(field, j, k)
        n, step, D = 1 << k, 1 << (dom.extended_k - k), j - 1
        a = synth.field_elements(rng.randrange(1 << 30), batch * n).reshape(batch, n, 4)
        d = torch.from_numpy(a.view(np.int64).copy()).cuda()
        full = dom.coeff_to_extended(d).cpu().numpy().view(np.uint64).reshape(batch, n * step, 4)
        nb = rng.choice([D, step, rng.randrange(1, step + 1)])
        blk = dom.coeff_to_extended_blocks(d, nb).cpu().numpy().view(np.uint64).reshape(batch, nb, n, 4)
        ok = all((blk[:, r] == full[:, r::step]).all() for r in range(nb))
        # a polynomial of degree < D n given by its values on D blocks comes back (no division)
        h = synth.field_elements(rng.randrange(1 << 30), D * n)
        pieces = torch.from_numpy(h.reshape(D, n, 4).view(np.int64).copy()).cuda()
        # values of h on block r = sum_i c_r^i * (block r of piece i): use the library's own block transform per piece, combined on the host ring
        vals = dom.coeff_to_extended_blocks(pieces, D)  # (D pieces, D blocks, n, 4): piece i on block r
        # the full-domain chain on h zero-padded is the cross-check: extended_to_coeff(coeff_to_extended) is only defined for n coefficients,
        # so compare blocks_to_quotient against the identity on ONE piece: h = piece 0 (degree < n): its blocks are vals[0]
        back = dom.blocks_to_quotient(vals[0].contiguous().clone(), divide_by_vanishing=False).cpu().numpy().view(np.uint64)
        ok = ok and (back[:n] == h[:n]).all() and (back[n:] == 0).all()
        if not ok:
            fails.append(which)
            print("BLOCKS MISMATCH", field, k, j, batch, nb, flush=True)
    elif which == "sparse":
        # round 4: a batch of commitments whose columns mix the witness's value classes with full-size ones in random order (sparse-column path,
        # dense runs of every length) against the same columns committed one at a time over an UNTABLED copy of the base set (plain windowed pipeline)
        curve = rng.choice(["pallas", "vesta"])
        sf = api.SCALAR_FIELD[curve]
        k = rng.randrange(12, 17)
        n = 1 << k
        b = rng.randrange(8, 25)
        seed_a, seed_b = rng.randrange(1, 1 << 40), rng.randrange(1, 1 << 30)
        bases = api.Bases.generate(curve, seed_a, seed_b, n + 1)
        try:
            bases.precompute(0)
        except api.TrhError:
            pass
        # the reference for the lone commitments: the same points WITHOUT tables -- a lone MSM over a tabled set asks the sampler as well
        # (the unit path asks lone commitments too) and would compare the unit path with itself; without tables it is the plain windowed pipeline
        plain = api.Bases.generate(curve, seed_a, seed_b, n + 1)
        cols = np.zeros((b, n, 4), dtype=np.uint64)
        live = max(1, n // rng.choice([1, 2, 4, 8]))
        # a third of the trials draw only from the classes whose sampled rows show nothing but 0 / 1: such a chunk takes the unit path (ones
        # summed from the table, tiny columns summed directly, the others -- "flagbig" with many scalars -- through the compact pipeline)
        palette = ["flag", "word", "byte", "full", "full", "zero", "ones", "hidden", "edge"] if rng.random() < 0.65 else ["flag", "flag", "flagbig", "zero", "ones", "hidden"]
        for i in range(b):
            kind = rng.choice(palette)
            raw = synth.splitmix64_stream(rng.randrange(1 << 30), 0, live)
            if kind == "flag":
                cols[i, :live, 0] = raw & np.uint64(1)
            elif kind == "word":
                cols[i, :live, 0] = raw & np.uint64((1 << 32) - 1)
            elif kind == "byte":
                cols[i, :live, 0] = raw & np.uint64(255)
            elif kind == "full":
                cols[i] = scalars(sf, n, "uniform")
            elif kind == "ones":
                cols[i, :, 0] = 1
            elif kind == "hidden":  # full-size everywhere except on the rows the sampler reads
                cols[i] = scalars(sf, n, "uniform")
                cols[i, :: max(1, n // 1024)] = 0
            elif kind == "edge":    # a fraction of full-size rows around the list capacity
                step = rng.choice([4, 8, 16, 32])
                cols[i, ::step] = scalars(sf, (n + step - 1) // step, "uniform")
            elif kind == "flagbig":  # flags plus full-size scalars on rows the sampler skips: few (tiny) or many (general on the unit path)
                cols[i, :live, 0] = raw & np.uint64(1)
            if kind in ("flag", "word", "byte", "ones", "flagbig"):  # small canonical integers -> Montgomery form, a few full-size blinding rows at the end
                d0 = torch.from_numpy(cols[i].view(np.int64)).cuda()
                api._check(api.lib().trh_field_op_dev(api.FIELD_ID[sf], api.FIELD_OPS["to_mont"], api._devptr(d0), None, api._devptr(d0), n, None))
                torch.cuda.synchronize()
                cols[i] = d0.cpu().numpy().view(np.uint64)
                if rng.random() < 0.7:
                    cols[i, n - 6:] = scalars(sf, 6, "uniform")
                if kind == "flagbig":
                    step = max(1, n // 1024)
                    cnt = rng.choice([1, 9, 10, 11, 12, 40, 200])
                    if step > 1:
                        rows = [step * rng.randrange(n // step) + 1 + rng.randrange(step - 1) for _ in range(cnt)]
                        cols[i, rows] = scalars(sf, cnt, "uniform")
        blinds = synth.field_elements(rng.randrange(1 << 30), b)
        d = torch.from_numpy(cols.view(np.int64)).cuda()
        got = bases.commit_batch_dev(d, n, b, blinds)
        ok = True
        for i in rng.sample(range(b), min(b, 6)):
            one = torch.from_numpy(np.concatenate([cols[i], blinds[i][None]]).view(np.int64)).cuda()
            ref = plain.msm_dev(one, n + 1)
            ok = ok and (ref == got[i]).all() and (bases.msm_dev(one, n + 1) == ref).all()  # batch == plain lone == tabled lone
        if cpu_ref is not None:
            i = rng.randrange(b)
            want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, np.concatenate([cols[i], blinds[i][None]]), bases.download(), threads=cpu_ref.hardware_threads()))
            ok = ok and (np.asarray(got[i])[:8] == want).all()
            _ORACLE_COUNT[0] += 1
        bases.destroy()
        plain.destroy()
        if not ok:
            fails.append(which)
            print("SPARSE MISMATCH", curve, k, b, flush=True)
    elif which == "padded":
        # round 4: trh_best_fft on zero-padded host vectors (slots that are zero throughout are not sent; probed padding is cleared at once and
        # verified later) with non-zero elements dropped at random places of the padding, against the resident transform of the same vector
        field = rng.choice(["fp", "fq"])
        k = rng.randrange(16, 22)
        n = 1 << k
        dom = poly.EvaluationDomain(field, 3, k)
        w = dom._w["omega"]
        a = np.zeros((n, 4), dtype=np.uint64)
        data = n >> rng.randrange(0, 5)
        a[:data] = synth.field_elements(rng.randrange(1 << 30), data)
        for _ in range(rng.choice([0, 0, 1, 3])):
            a[rng.randrange(n)] = synth.field_elements(rng.randrange(1 << 30), 1)[0]
        d = torch.from_numpy(a.view(np.int64).copy()).cuda()
        api.ntt_dev(field, d, k, w)
        torch.cuda.synchronize()
        work = a.copy()
        api.best_fft_inplace(field, work, w, k)
        ok = (work == d.cpu().numpy().view(np.uint64)).all()
        if cpu_ref is not None and k <= 18:
            ok = ok and (work == cpu_ref.best_fft(field, a, np.asarray(w, dtype=np.uint64), k, threads=cpu_ref.hardware_threads())).all()
            _ORACLE_COUNT[0] += 1
        if not ok:
            fails.append(which)
            print("PADDED MISMATCH", field, k, data, flush=True)
    elif which == "opening":
        # round 6: the IPA opening over a tabled g || w || u (k <= 13: msm_small_kernel from round 0 on level 0 of the table; k >= 14: k - 12 full-size
        # fixed-base rounds, the generator collapse of ipafold.hip, the other rounds over 2^12 + 2 points) against the same opening without tables
        # (per-window MSMs over the original generators in every round): the same transcript, item for item
        import hashlib
        from tiny_ram_halo2_amd import ipa
        curve = rng.choice(["pallas", "vesta"])
        sf = api.SCALAR_FIELD[curve]
        k = rng.choice([rng.randrange(1, 9), rng.randrange(9, 14), rng.randrange(14, 17)])
        n = 1 << k
        g = api.Bases.generate(curve, rng.randrange(1, 1 << 40), rng.randrange(1, 1 << 30), n + 1).download()
        u = api.Bases.generate(curve, rng.randrange(1, 1 << 40), 1, 1).download()
        p_h = scalars(sf, n, rng.choice(["uniform", "uniform", "small", "edge"]))
        s_h = scalars(sf, n, "uniform")
        m = MOD[sf]
        p_blind, s_blind, x3 = rng.randrange(m), rng.randrange(m), rng.randrange(1, m)
        draws = [rng.randrange(1, m) for _ in range(2 * k)]

        class Tr:
            def __init__(self):
                self.h, self.items = hashlib.blake2b(b"soak"), []
            def write_point(self, jac):
                b = np.ascontiguousarray(jac, dtype=np.uint64)[:8].tobytes(); self.h.update(b"P" + b); self.items.append(b)
            def write_scalar(self, l):
                b = np.ascontiguousarray(l, dtype=np.uint64).tobytes(); self.h.update(b"S" + b); self.items.append(b)
            def squeeze_challenge_scalar(self):
                self.h.update(b"C")
                return int.from_bytes(self.h.digest()[:40], "little") % (m - 1) + 1

        outs = []
        for tables in (True, False):
            params = poly.Params(curve, k, g[:n], g[:n], g[n:n + 1], u=u, precompute=tables)
            it = iter(draws)
            tr = Tr()
            cf = ipa.create_proof_native(params, lambda: next(it), tr, torch.from_numpy(p_h.view(np.int64)).cuda(), p_blind, x3, s_h, s_blind)
            outs.append((cf, tr.items))
        if outs[0] != outs[1]:
            fails.append(which)
            print("OPENING MISMATCH", curve, k, flush=True)
    elif which == "sharded":
        # round 4: a base set range-sharded over the device group: host scalars (one uploader thread per shard), page-locked host scalars,
        # device scalars with and without the forced no-peer hand-over, random sub-ranges -- against the same MSM on one context
        if api.group_size() != GROUP:
            try:
                api.init_multi([0] * GROUP)
            except api.TrhError:
                return  # another group shape is active in this process: nothing to test here
        curve = rng.choice(["pallas", "vesta"])
        sf = api.SCALAR_FIELD[curve]
        n = rng.randrange(1 << 13, 1 << 18)
        s0, dd = rng.randrange(1, 1 << 40), rng.randrange(1, 1 << 30)
        sc = scalars(sf, n, rng.choice(["uniform", "uniform", "small", "edge"]))
        api.set_shard_min(1 << 62)
        one = api.Bases.generate(curve, s0, dd, n)
        api.set_shard_min(1 << 12)
        try:
            sh = api.Bases.generate(curve, s0, dd, n)
            lo = rng.randrange(0, n - 1)
            cnt = rng.randrange(1, n - lo + 1)
            if rng.random() < 0.5:
                lo, cnt = 0, n
            part = np.ascontiguousarray(sc[lo:lo + cnt])
            want = one.msm(part, offset=lo)
            ok = (sh.msm(part, offset=lo) == want).all()
            pinned = torch.from_numpy(part.view(np.int64)).pin_memory()
            ok = ok and (sh.msm(pinned.numpy().view(np.uint64), offset=lo) == want).all()
            dsc = torch.from_numpy(part.view(np.int64)).cuda()
            ok = ok and (sh.msm_dev(dsc, cnt, offset=lo) == want).all()
            # (the forced no-peer hand-over is an option fixed per process since round 6: tests/test_gpu_multi.py runs it in a child)
            sh.destroy()
        finally:
            api.set_shard_min(1 << 62)
        one.destroy()
        if not ok:
            fails.append(which)
            print("SHARDED MISMATCH", curve, n, lo, cnt, flush=True)
    elif which == "hostio":
        # host-pointer entries against the device-resident ones (round 3): batch FFT, batched commitments, range-tiled host MSMs
        field = rng.choice(["fp", "fq"])
        k = rng.randrange(1, 19)
        count = rng.randrange(1, 9)
        m = MOD[field]
        dom = poly.EvaluationDomain(field, 3, k)
        w = dom._w["omega"]
        cols = [np.ascontiguousarray(synth.field_elements(rng.randrange(1 << 30), 1 << k)) for _ in range(count)]
        single = [api.best_fft(field, c, w, k) for c in cols]
        api.best_fft_batch(field, cols, w, k)
        ok = all((a == b).all() for a, b in zip(cols, single))
        curve = rng.choice(["pallas", "vesta"])
        n = rng.randrange(1, 1 << 15)
        bases = api.Bases.generate(curve, rng.randrange(1, 1 << 40), rng.randrange(1, 1 << 30), n + 1)
        if rng.random() < 0.5:
            try:
                bases.precompute(0)
            except api.TrhError:
                pass
        batch = rng.randrange(1, 40)
        sf = api.SCALAR_FIELD[curve]
        polys = [np.ascontiguousarray(scalars(sf, n, rng.choice(["uniform", "small", "edge"]))) for _ in range(batch)]
        blinds = synth.field_elements(rng.randrange(1 << 30), batch)
        got = bases.commit_batch_host(polys, blinds)
        pick = rng.randrange(batch)
        ok = ok and (got[pick] == bases.msm(np.concatenate([polys[pick], blinds[pick][None]]))).all()
        xy = bases.download()
        sc = np.concatenate([polys[0], blinds[0][None]])
        ok = ok and (api.best_multiexp(curve, sc, xy) == got[0]).all() and (bases.msm(sc) == got[0]).all()
        if not ok:
            fails.append(which)
            print("HOSTIO MISMATCH", field, k, count, curve, n, batch, flush=True)
    elif which == "products":
        # product columns from rows of factors (trh_product_terms_dev + batch_invert_mul + batched prefix product) against big integers,
        # and ff::BatchInvert alone at an arbitrary length (a * a^-1 == 1 wherever a != 0, zeros left alone)
        field = rng.choice(["fp", "fq"])
        m = MOD[field]
        k = rng.randrange(1, 13)
        n = 1 << k
        rinv = pow(1 << 256, -1, m)
        pool_h = [synth.field_elements(rng.randrange(1 << 30), n) for _ in range(5)]
        pool_d = [torch.from_numpy(c.view(np.int64)).cuda() for c in pool_h]
        pool_i = [[int.from_bytes(r.tobytes(), "little") * rinv % m for r in c] for c in pool_h]
        def row():
            terms_d, terms_i = [], []
            for _ in range(rng.randrange(1, 5)):
                x, g = rng.randrange(5), rng.randrange(m)
                y = rng.randrange(5) if rng.random() < 0.6 else None
                c = rng.randrange(m)
                terms_d.append((pool_d[x], pool_d[y] if y is not None else None, limbs(field, c) if y is not None else None, limbs(field, g)))
                terms_i.append((x, y, c, g))
            return terms_d, terms_i
        rows = [(row(), row()) for _ in range(rng.randrange(1, 5))]
        z = permutation.grand_products_terms(field, k, [r[0][0] for r in rows], [r[1][0] for r in rows]).cpu().numpy().view(np.uint64)
        ok = True
        val = lambda terms, i: __import__("functools").reduce(lambda a, t: a * ((pool_i[t[0]][i] + (t[2] * pool_i[t[1]][i] if t[1] is not None else 0) + t[3]) % m) % m, terms, 1)
        for r, (nr, dr) in enumerate(rows):
            acc = 1
            for i in range(n):
                if int.from_bytes(z[r][i].tobytes(), "little") * rinv % m != acc:
                    ok = False
                    break
                d = val(dr[1], i)
                acc = acc * val(nr[1], i) * (pow(d, -1, m) if d else 0) % m
        cnt = rng.choice([rng.randrange(1, 300), rng.randrange(300, 1 << 15), rng.randrange(1 << 15, 1 << 18)])
        a = synth.field_elements(rng.randrange(1 << 30), cnt).copy()
        for zi in range(0, cnt, rng.randrange(50, 5000)):
            a[zi] = 0
        da = torch.from_numpy(a.view(np.int64)).cuda()
        inv = da.clone()
        api.batch_invert_dev(field, inv, cnt)
        prod = torch.empty_like(da)
        api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS["mul"], api._devptr(da), api._devptr(inv), api._devptr(prod), cnt, None))
        torch.cuda.synchronize()
        ph, ih = prod.cpu().numpy().view(np.uint64), inv.cpu().numpy().view(np.uint64)
        zero = ~a.any(axis=1)
        one = np.array(limbs(field, 1), dtype=np.uint64)
        ok = ok and (ph[~zero] == one).all() and not ih[zero].any()
        if not ok:
            fails.append(which)
            print("PRODUCTS MISMATCH", field, k, cnt, flush=True)
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
        if n <= 4096:  # A' is in the order of the canonical integers
            rinv = pow(1 << 256, -1, m)
            ints = [int.from_bytes(r.tobytes(), "little") * rinv % m for r in pa_h]
            ok = ok and ints == sorted(ints)
        if not ok:
            fails.append(which)
            print("LOOKUP MISMATCH", field, n, tsize, flush=True)


def run(budget_s: float, seed: int, oracle=None, kinds=None):
    """random trials for budget_s seconds from `seed` (logged); returns (stats, failures)"""
    global rng, cpu_ref
    rng = random.Random(seed)
    cpu_ref = oracle
    api.init(0)
    stats = {k: 0 for k in dict.fromkeys(KINDS)}
    stats["vs_oracle"] = 0
    fails = []
    t_end = time.time() + budget_s
    print(f"soak: seed {seed}, budget {budget_s:.0f} s, oracle {'yes' if oracle is not None else 'no'}", flush=True)
    while time.time() < t_end:
        which = rng.choice(kinds or KINDS)
        before = _ORACLE_COUNT[0]
        trial(which, fails)
        stats[which] += 1
        stats["vs_oracle"] += _ORACLE_COUNT[0] - before
    return stats, fails


_ORACLE_COUNT = [0]

if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    stats, fails = run(budget, seed)
    print("soak:", stats, "failures:", len(fails))
    sys.exit(1 if fails else 0)
