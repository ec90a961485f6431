"""Throughput of batched commit-size MSMs: tools/batch_probe.py [log_n] [batch] [window_bits] [fixed-base window bits, -1 = off]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, synth
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cbits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = (1 << log_n) + 1
api.init(0)
bases = api.Bases.generate("vesta", synth.BASE_S0, synth.BASE_D, n)
sc = torch.from_numpy(synth.field_elements(0x78, n * batch).view(np.int64)).cuda()
st = torch.cuda.current_stream().cuda_stream
api.set_window_bits(cbits)
fbits = int(sys.argv[4]) if len(sys.argv) > 4 else -1
if fbits >= 0:
    torch.cuda.synchronize(); t = time.perf_counter()
    used = bases.precompute(fbits)
    print(f"precompute c={used}: {(time.perf_counter() - t) * 1e3:.1f} ms")
bases.msm_batch_dev(sc, n, batch, stream=st)
torch.cuda.synchronize(); t = time.perf_counter()
reps = 3
for _ in range(reps):
    bases.msm_batch_dev(sc, n, batch, stream=st)
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / reps * 1e3
print(f"batch {batch} x MSM(2^{log_n}+1) c={cbits}: {ms:.2f} ms total, {ms / batch:.3f} ms per MSM, {n * batch / ms / 1e3:.1f} Mpairs/s")
api.set_timing(True)
bases.msm_batch_dev(sc, n, batch, stream=st)
print({k: round(v, 3) for k, v in api.last_timing().items()})
api.set_timing(False)
