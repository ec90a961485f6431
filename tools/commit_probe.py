"""One batch of commit_lagrange over witness-shaped columns of ONE value class: tools/commit_probe.py [kind] [blinded 0/1] [batch] [k]
kind: flag | word | even | sorted | full (tiny-ram-halo2_amd/replay.py::witness_columns)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, poly, replay, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "flag"
blinded = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
k = int(sys.argv[4]) if len(sys.argv) > 4 else 18
n = 1 << k
api.init(0)
curve, field = "vesta", "fp"
gl = api.Bases.generate(curve, synth.BASE_S0 + 77, synth.BASE_D + 2, n + 1)
gl.precompute(int(os.environ.get("TRH_PROBE_TABLE_C", "0")))
can = replay.witness_columns(kind, blinded, 0xC01, batch, n, 2 * (k - 2))
d = torch.from_numpy(can.view(np.int64)).cuda()
st = torch.cuda.current_stream().cuda_stream
if kind != "full":
    api._check(api.lib().trh_field_op_dev(api.FIELD_ID[field], api.FIELD_OPS["to_mont"], api._devptr(d), None, api._devptr(d), batch * n, st))
blinds = synth.field_elements(7, batch)
for _ in range(2):
    gl.commit_batch_dev(d, n, batch, blinds, stream=st)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    gl.commit_batch_dev(d, n, batch, blinds, stream=st)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
print(f"commit_lagrange batch of {batch} '{kind}' columns (blinded={blinded}), k={k}: {ms:.3f} ms = {ms / batch * 1e3:.1f} us per column")
