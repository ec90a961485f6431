"""Timing of the grand-product columns (permutation argument): tools/product_probe.py [k]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, permutation, synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
api.init(0)
n = 1 << k
cols = [torch.from_numpy(synth.field_elements(0x90 + j, n).view(np.int64)).cuda() for j in range(8)]
pc = permutation.ProductColumn("fp", k, 4)
pc.compute(cols[:4], cols[4:], 123456789, 987654321)
torch.cuda.synchronize(); t = time.perf_counter()
reps = 10
for _ in range(reps):
    pc.compute(cols[:4], cols[4:], 123456789, 987654321)
torch.cuda.synchronize()
print(f"permutation product column, n=2^{k}, 4 columns: {(time.perf_counter() - t) / reps * 1e3:.3f} ms")
st = torch.cuda.current_stream().cuda_stream
for name, fn in (("batch_invert", lambda: api.batch_invert_dev("fp", cols[0].clone(), n, stream=st)),
                 ("prefix_product", lambda: api.prefix_product_dev("fp", cols[0], cols[1], n, stream=st))):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"  {name}: {(time.perf_counter() - t) / reps * 1e3:.3f} ms")
