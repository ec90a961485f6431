"""Timeline of ONE trh_msm over a witness column in host memory (option trace = 1 prints the host-pointer entry's steps to stderr):
TRH_TRACE=1 python3 tools/exp/lone_host_trace.py [kind]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, replay, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "word"
api.init(0)
k = 18; n = 1 << k
bases = api.Bases.generate("vesta", synth.BASE_S0, synth.BASE_D, n + 1)
bases.precompute(0)
can = replay.witness_columns(kind, True, 7, 1, n, 32)
d = torch.from_numpy(can.view(np.int64)).cuda()
api._check(api.lib().trh_field_op_dev(api.FIELD_ID["fp"], api.FIELD_OPS["to_mont"], api._devptr(d), None, api._devptr(d), n, None))
torch.cuda.synchronize()
sc = np.concatenate([d[0].cpu().numpy().view(np.uint64), synth.field_elements(3, 1)])
for _ in range(4): bases.msm(sc)
sys.stderr.write("==== traced call\n")
t0 = time.perf_counter(); bases.msm(sc); print(f"{kind}: {(time.perf_counter() - t0) * 1e3:.3f} ms")
