"""Timing of the real generator collapse (trh_bases_fold_dev) and of batch-2 MSMs at the shrinking sizes of the IPA rounds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, synth
api.init(0)
curve = "vesta"
k = 18
n = 1 << k
g = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
xy = torch.from_numpy(g.download().view(np.int64)).cuda()
u = synth.field_elements(0x33, 1).reshape(4)
sc = torch.from_numpy(synth.field_elements(0x1FA, 2 * n).view(np.int64)).cuda()
def tm(f, reps=3):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for lh in range(17, 5, -2):
    half = 1 << lh
    w = xy.clone()
    t_fold = tm(lambda: api.bases_fold_dev(curve, w, w[half:], half, u))
    b = api.Bases.wrap_device(curve, w.data_ptr(), half)
    t_msm = tm(lambda: b.msm_batch_dev(sc, half, 2))
    print(f"half=2^{lh}: fold {t_fold:.3f} ms   batch-2 msm {t_msm:.3f} ms")
t_msm = tm(lambda: g.msm_batch_dev(sc, n, 2))
print(f"n=2^{k}: batch-2 msm over resident bases {t_msm:.3f} ms")
