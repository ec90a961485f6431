"""kernel timeline of ONE lone commitment (trh_msm_dev, 2^18 + 1 scalars, tables) per witness class: tools/lone_trace.py  (run under tools/prof_cmd.sh)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, replay, synth
api.init(0)
k = 18; n = 1 << k
bases = api.Bases.generate("vesta", synth.BASE_S0, synth.BASE_D, n + 1)
bases.precompute(0)
for kind in ("flag", "word", "full"):
    can = replay.witness_columns(kind, True, 7, 1, n, 32)
    d = torch.from_numpy(np.concatenate([can[0], synth.field_elements(3, 1)]).view(np.int64)).cuda()
    api._check(api.lib().trh_field_op_dev(api.FIELD_ID["fp"], api.FIELD_OPS["to_mont"], api._devptr(d), None, api._devptr(d), n, None))
    torch.cuda.synchronize()
    for _ in range(3):
        bases.msm_dev(d, n + 1)
    torch.cuda.synchronize()
