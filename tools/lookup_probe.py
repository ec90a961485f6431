"""Timing of the lookup argument's permuted columns: tools/lookup_probe.py [log_n] [table size log]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, permutation, synth
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
log_t = int(sys.argv[2]) if len(sys.argv) > 2 else 16
api.init(0)
n = 1 << log_n
distinct = torch.from_numpy(synth.field_elements(0x7AB, 1 << log_t).view(np.int64)).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
table = distinct[torch.arange(n, device="cuda") % (1 << log_t)].contiguous()
inp = distinct[torch.randint(0, 1 << log_t, (n,), device="cuda", generator=g)].contiguous()
permutation.lookup_permute("fp", inp, table)
torch.cuda.synchronize(); t = time.perf_counter()
reps = 5
for _ in range(reps):
    permutation.lookup_permute("fp", inp, table)
torch.cuda.synchronize()
print(f"lookup_permute n=2^{log_n}, {1 << log_t} distinct table values: {(time.perf_counter() - t) / reps * 1e3:.3f} ms")
