"""Commit-size MSM on witness-like scalars: tools/skew_probe.py [log_n]
(TinyRAM advice columns: only n/4 rows used, values are flags or small words)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, synth, poly
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = (1 << log_n) + 1
api.init(0)
bases = api.Bases.generate("vesta", synth.BASE_S0, synth.BASE_D, n)
st = torch.cuda.current_stream().cuda_stream
P = poly._MODULUS["fp"]; R = (1 << 256) % P
def mont_small(vals):
    out = np.zeros((len(vals), 4), np.uint64)
    for i, v in enumerate(np.unique(vals)):
        pass
    table = {int(v): [(int(v) * R % P >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for v in np.unique(vals)}
    lut = np.array([table[int(v)] for v in np.unique(vals)], np.uint64)
    idx = np.searchsorted(np.unique(vals), vals)
    return lut[idx]
rng = np.random.default_rng(1)
cases = {
    "random 254-bit": synth.field_elements(0x99, n),
    "flags {0,1}, n/4 rows": mont_small(np.where(np.arange(n) < n // 4, rng.integers(0, 2, n), 0)),
    "8-bit words, n/4 rows": mont_small(np.where(np.arange(n) < n // 4, rng.integers(0, 256, n), 0)),
    "16-bit words, all rows": mont_small(rng.integers(0, 65536, n)),
    "all ones": mont_small(np.ones(n, np.int64)),
}
for name, sc in cases.items():
    d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
    bases.msm_dev(d, n, stream=st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        bases.msm_dev(d, n, stream=st)
    torch.cuda.synchronize()
    print(f"{name:26s} {(time.perf_counter() - t) / 5 * 1e3:8.3f} ms")
