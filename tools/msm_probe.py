"""Phase timing of one MSM: tools/msm_probe.py [log_n] [curve] [window_bits] [plus]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, synth
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
curve = sys.argv[2] if len(sys.argv) > 2 else "vesta"
cbits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = (1 << log_n) + (int(sys.argv[4]) if len(sys.argv) > 4 else 1)
api.init(0)
bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
sc = torch.from_numpy(synth.field_elements(0x77, n).view(np.int64)).cuda()
st = torch.cuda.current_stream().cuda_stream
api.set_window_bits(cbits)
for _ in range(2):
    res = bases.msm_dev(sc, n, stream=st)
api.set_timing(True)
acc = {}
reps = 5
for _ in range(reps):
    bases.msm_dev(sc, n, stream=st)
    for k, v in api.last_timing().items():
        acc[k] = acc.get(k, 0) + v / reps
api.set_timing(False)
import time
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps):
    bases.msm_dev(sc, n, stream=st)
torch.cuda.synchronize(); wall = (time.perf_counter() - t) / reps * 1e3
res2 = bases.msm_dev(sc, n, stream=st)
assert (res == res2).all()
print(f"n={n} {curve}", {k: round(v, 3) for k, v in acc.items()}, f"wall {wall:.3f} ms  {n / wall / 1e3:.1f} Mpairs/s  point {int(res[0]):016x}{int(res[4]):016x}")
