"""where the grand-product columns' time goes (k = 18, 47 permutation chunks + 31 lookup products): tools/products_probe.py [k]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, expr, permutation, poly, synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << k
api.init(0)
dev = torch.device("cuda:0")
field = "fp"
m_ = poly._MODULUS[field]
beta, gamma = 0xBE7A % m_, 0x6A33A % m_
wit = [torch.from_numpy(synth.field_elements(0x9E0 + j, n).view(np.int64)).to(dev) for j in range(8)]
om = torch.empty((n, 4), dtype=torch.int64, device=dev)
api.powers_dev(field, om, n, expr._limbs(field, permutation.omega(field, k)))
NP, NL = 47, 31
pcs = [permutation.ProductColumn(field, k, 4, first_column=4 * c) for c in range(NP)]
evs = [pc.evaluator(beta, gamma) for pc in pcs]
sets = [pc.columns(wit[:4], wit[4:], om) for pc in pcs]
lk = permutation.lookup_product(field, k, beta, gamma)
evs += [lk.ev] * NL
sets += [{("advice", i): wit[(i + li) % 8] for i in range(4)} for li in range(NL)]
st = torch.cuda.current_stream().cuda_stream


def timed(f, reps=3):
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    e0.record()
    for _ in range(reps):
        r = f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t) / reps * 1e3, r


ms, wall, _ = timed(lambda: permutation.grand_products_batch(field, k, evs, sets))
print(f"grand_products_batch x{len(evs)}: {ms:.3f} ms (wall {wall:.3f})")
ms, wall, _ = timed(lambda: [ev.eval(c, k, 1, stream=st) for ev, c in zip(evs, sets)])
print(f"  the {len(evs)} evaluator calls: {ms:.3f} ms (wall {wall:.3f})")
ms, wall, _ = timed(lambda: evs[0].eval(sets[0], k, 1, stream=st), reps=20)
print(f"  one permutation-chunk evaluator call: {ms * 1e3:.1f} us (wall {wall * 1e3:.1f})")
ms, wall, _ = timed(lambda: evs[-1].eval(sets[-1], k, 1, stream=st), reps=20)
print(f"  one lookup-product evaluator call: {ms * 1e3:.1f} us (wall {wall * 1e3:.1f})")
rows = len(evs)
num = torch.from_numpy(synth.field_elements(5, n).view(np.int64)).to(dev).repeat(rows, 1, 1).contiguous()
den = num.clone()
ms, wall, _ = timed(lambda: api.batch_invert_dev(field, den, rows * n, stream=st))
print(f"  batch inversion of {rows} x 2^{k}: {ms:.3f} ms")
ms, wall, _ = timed(lambda: api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS['mul'], api._devptr(num), api._devptr(den), api._devptr(den), rows * n, st)))
print(f"  multiply: {ms:.3f} ms")
z = torch.empty_like(num)
ms, wall, _ = timed(lambda: api._check(api.lib().trh_field_prefix_product_rows_dev(api.FIELD_ID[field], api._devptr(den), api._devptr(z), n, rows, st)))
print(f"  prefix products: {ms:.3f} ms")
num_rows, den_rows = [], []
for c in range(NP):
    nr, dr = permutation.permutation_terms(field, wit[:4], wit[4:], om, beta, gamma, first_column=4 * c)
    num_rows.append(nr); den_rows.append(dr)
for li in range(NL):
    nr, dr = permutation.lookup_terms(field, *[wit[(i + li) % 8] for i in range(4)], beta, gamma)
    num_rows.append(nr); den_rows.append(dr)
ms, wall, zt = timed(lambda: permutation.grand_products_terms(field, k, num_rows, den_rows))
print(f"grand_products_terms x{len(num_rows)}: {ms:.3f} ms (wall {wall:.3f})")
zb = permutation.grand_products_batch(field, k, evs, sets)
torch.cuda.synchronize()
print("  same z columns as the expression-program path:", bool((zt == zb).all()))
nd = torch.empty((2 * rows, n, 4), dtype=torch.int64, device=dev)
ms, wall, _ = timed(lambda: api.product_terms_dev(field, num_rows + den_rows, n, nd, stream=st))
print(f"  product_terms: {ms:.3f} ms")
ms, wall, _ = timed(lambda: api.batch_invert_mul_dev(field, nd[rows:], nd[:rows], rows * n, stream=st))
print(f"  batch_invert_mul: {ms:.3f} ms")
