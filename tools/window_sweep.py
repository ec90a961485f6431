"""Sweep the MSM window width per size: prints wall ms per (log_n, c)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, synth
api.init(0)
st = torch.cuda.current_stream().cuda_stream
for log_n in [int(x) for x in sys.argv[1].split(",")]:
    n = (1 << log_n) + 1
    bases = api.Bases.generate("vesta", synth.BASE_S0, synth.BASE_D, n)
    sc = torch.from_numpy(synth.field_elements(0x79, n).view(np.int64)).cuda()
    row = []
    for c in range(max(3, log_n - 9), min(18, log_n + 1) + 1):
        api.set_window_bits(c)
        bases.msm_dev(sc, n, stream=st)
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = 5 if log_n < 22 else 2
        for _ in range(reps):
            bases.msm_dev(sc, n, stream=st)
        torch.cuda.synchronize()
        row.append((c, round((time.perf_counter() - t) / reps * 1e3, 3)))
    best = min(row, key=lambda r: r[1])
    print(f"2^{log_n}: best c={best[0]} {best[1]} ms |", " ".join(f"{c}:{ms}" for c, ms in row))
