"""coeff_to_extended alone (zero-pad + coset shift fused into the first NTT pass): tools/ext_probe.py [k] [batch] [j] [blocks]
(blocks > 0: the coset-block form with that many blocks per column)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, poly, synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 28
api.init(0)
dom = poly.EvaluationDomain("fp", int(sys.argv[3]) if len(sys.argv) > 3 else 6, k)
n = 1 << k
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 0
run = (lambda x: dom.coeff_to_extended_blocks(x, blocks)) if blocks else dom.coeff_to_extended
d = torch.from_numpy(synth.field_elements(11, n * batch).view(np.int64).copy()).cuda().reshape(batch, n, 4)
for _ in range(2):
    ext = run(d)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
reps = 5
for _ in range(reps):
    ext = run(d)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"coeff_to_extended k={k} -> {dom.extended_k} x{batch} blocks={blocks}: {ms:.3f} ms, {ms / batch * 1e3:.1f} us per column")
