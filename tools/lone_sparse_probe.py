"""A lone commitment (trh_msm: n + 1 host scalars over a resident set with tables) of one witness column per value class, with and without the
sparse-column path (TRH_SPARSE is read once per process: run twice).  tools/lone_sparse_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from tiny_ram_halo2_amd import api, replay, synth
api.init(0)
k = 18; n = 1 << k
bases = api.Bases.generate("vesta", synth.BASE_S0, synth.BASE_D, n + 1)
bases.precompute(0)
out = []
for kind in ("flag", "word", "even", "sorted", "full"):
    can = replay.witness_columns(kind, True, 7, 1, n, 32)
    d = torch.from_numpy(can.view(np.int64)).cuda()
    api._check(api.lib().trh_field_op_dev(api.FIELD_ID["fp"], api.FIELD_OPS["to_mont"], api._devptr(d), None, api._devptr(d), n, None))
    torch.cuda.synchronize()
    col = d[0].cpu().numpy().view(np.uint64)
    sc = np.concatenate([col, synth.field_elements(3, 1)])
    dsc = torch.from_numpy(sc.view(np.int64)).cuda()
    for _ in range(3): bases.msm(sc); bases.msm_dev(dsc, n + 1)
    t0 = time.perf_counter()
    for _ in range(20): bases.msm(sc)
    th = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for _ in range(20): bases.msm_dev(dsc, n + 1)
    td = (time.perf_counter() - t0) / 20 * 1e3
    out.append(f"{kind}: host scalars {th:.3f} ms, device scalars {td:.3f} ms")
print(f"TRH_SPARSE={os.environ.get('TRH_SPARSE', '1')}: " + "; ".join(out))
