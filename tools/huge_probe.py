"""MSMs beyond the swept sizes (2^27 .. 2^30 pairs): tools/huge_probe.py [log_n] [curve] [extra pairs].  The scalar vector is a 2^20 block
repeated, so the closed-form expectation (bases with known logs s0 + i d) needs host arithmetic over one block only:
sum_i s_i (s0 + i d) = reps (s0 T0 + d T1) + d 2^20 T0 reps (reps - 1) / 2,  T0 = sum t_j,  T1 = sum j t_j."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, synth
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
curve = sys.argv[2] if len(sys.argv) > 2 else "pallas"
MOD = {"pallas": 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001, "vesta": 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001}
q = MOD[curve]
R = (1 << 256) % q
BL = 20
api.init(0)
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
assert 0 <= extra < (1 << BL)
n, reps = (1 << log_n) + extra, 1 << (log_n - BL)
block = synth.field_elements(0xB16 + log_n, 1 << BL)  # canonical values below 2^254: valid scalars of both curves
T0 = synth.weighted_scalar_sum(block, 1, 0)
T1 = synth.weighted_scalar_sum(block, 0, 1)
total = reps * (synth.BASE_S0 * T0 + synth.BASE_D * T1) + synth.BASE_D * (1 << BL) * T0 * (reps * (reps - 1) // 2)
if extra:
    total += synth.weighted_scalar_sum(block[:extra], synth.BASE_S0, synth.BASE_D, start=reps << BL)
total %= q
g1 = api.Bases.generate(curve, 1, 0, 1)
want = g1.msm(synth.ints_to_limbs([total * R % q]))
t = time.perf_counter()
bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
torch.cuda.synchronize()
print(f"bases: {(time.perf_counter() - t):.2f} s", flush=True)
d = torch.from_numpy(block.view(np.int64)).cuda().repeat(reps + 1, 1)[:n].contiguous()
torch.cuda.synchronize()
for it in range(2):
    t = time.perf_counter()
    got = bases.msm_dev(d, n, montgomery=False)
    ms = (time.perf_counter() - t) * 1e3
    print(f"msm {curve} n=2^{log_n}+{extra}: {'ok' if (got == want).all() else 'MISMATCH'} ({ms:.1f} ms, {n / ms / 1e3:.1f} Mpairs/s)", flush=True)
