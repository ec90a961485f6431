"""IPA opening alone at size k: tools/ipa_probe.py [k] [window bits override]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, ipa, poly, replay, synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << k
api.init(0)
if len(sys.argv) > 2:
    api.set_window_bits(int(sys.argv[2]))
curve = "vesta"
g = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n + 1)
params = poly.Params.__new__(poly.Params)
params.curve, params.k, params.n = curve, k, n
params._g = g
params.w = g.download(n, 1)
params.u = api.Bases.generate(curve, 4242, 1, 1).download()
if len(sys.argv) <= 2 and os.environ.get("TRH_IPA_TABLES", "1") != "0":
    g.precompute(0)  # Params use tables: ipa_bases() then builds g || w || u with its own table
m = poly._MODULUS["fp"]
p_dev = torch.from_numpy(synth.field_elements(0x1FA, n).view(np.int64)).cuda()
s_h = synth.field_elements(0x5A, n)
for rep in range(2):
    draws = iter(range(7, 10 ** 9, 13))
    torch.cuda.synchronize(); t = time.perf_counter()
    ipa.create_proof_native(params, lambda: next(draws), replay._FixedTranscript(m), p_dev, 0x1234, 0x77777, s_h, 0x99)
    torch.cuda.synchronize()
    print(f"ipa k={k}: {(time.perf_counter() - t) * 1e3:.1f} ms")
