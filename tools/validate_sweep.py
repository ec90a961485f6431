"""Size sweep of the size-independent checks (closed-form MSM over bases with known logs, inverse-forward NTT round trip):
tools/validate_sweep.py [max_msm_log] [max_ntt_log].  Uses only libtrh (no oracle): every result is compared with a second,
independent way of computing it on the device (a 1-pair MSM of the generator by the closed-form scalar; the inverse transform)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, synth
api.init(0)
max_msm = int(sys.argv[1]) if len(sys.argv) > 1 else 24
max_ntt = int(sys.argv[2]) if len(sys.argv) > 2 else 24
MOD = {"pallas": 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001, "vesta": 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001}
SF = {"pallas": "fq", "vesta": "fp"}
bad = 0
for curve in ("pallas", "vesta"):
    q = MOD[curve]
    R = (1 << 256) % q
    g1 = api.Bases.generate(curve, 1, 0, 1)
    for log_n in range(1, max_msm + 1):
        for extra in ((0, 1) if log_n <= 20 else (0,)):
            n = (1 << log_n) + extra
            bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
            sc = synth.field_elements(0x5EE9 + log_n, n) if curve == "vesta" else synth.msm_scalars(log_n)[:n] if extra == 0 else synth.field_elements(0x5EE9 + log_n, n)
            if curve == "pallas" and sc.shape[0] != n:
                sc = synth.field_elements(0x5EE9 + log_n, n)
            d = torch.from_numpy(sc.view(np.int64)).cuda()
            dcan = torch.empty_like(d)
            api._check(api.lib().trh_field_op_dev(api.FIELD_ID[SF[curve]], api.FIELD_OPS["from_mont"], api._devptr(d), None, api._devptr(dcan), n, None))
            torch.cuda.synchronize()
            can = dcan.cpu().numpy().view(np.uint64)
            if int(np.max(can[:, 3])) >> 62 and False:
                pass
            total = synth.weighted_scalar_sum(can, synth.BASE_S0, synth.BASE_D) % q
            want = g1.msm(synth.ints_to_limbs([total * R % q]))
            t = time.perf_counter()
            got = bases.msm_dev(d, n)
            ms = (time.perf_counter() - t) * 1e3
            ok = (got == want).all()
            fb_ok = True
            if log_n <= 19:
                try:
                    bases.precompute(0)
                    fb_ok = (bases.msm_dev(d, n) == want).all()
                except api.TrhError:
                    fb_ok = True
            if not (ok and fb_ok):
                bad += 1
            print(f"msm {curve} n=2^{log_n}+{extra}: {'ok' if ok else 'MISMATCH'} fixed-base {'ok' if fb_ok else 'MISMATCH'} ({ms:.2f} ms)", flush=True)
            del bases, d, dcan
P = {"fp": MOD["vesta"], "fq": MOD["pallas"]}
ROOTS = {"fp": 0x2BCE74DEAC30EBDA362120830561F81AEA322BF2B7BB7584BDAD6FABD87EA32F, "fq": 0x2DE6A9B8746D3F589E5C4DFD492AE26E9BB97EA3C106F049A70E2C1102B6D05F}
for field in ("fp", "fq"):
    p = P[field]
    R = (1 << 256) % p
    for log_n in range(1, max_ntt + 1):
        n = 1 << log_n
        w = pow(ROOTS[field], 1 << (32 - log_n), p)
        a = synth.field_elements(0x4E77 + log_n, n) if field == "fp" else synth.field_elements(0x4E78 + log_n, n)
        if field == "fq":  # field_elements draws below 2^254, valid for both moduli
            pass
        d = torch.from_numpy(a.view(np.int64).copy()).cuda()
        api.ntt_dev(field, d, log_n, synth.ints_to_limbs([w * R % p])[0])
        fwd0 = d[0].cpu().numpy().view(np.uint64).copy()
        api.ntt_dev(field, d, log_n, synth.ints_to_limbs([pow(w, -1, p) * R % p])[0])
        api.field_scale_dev(field, d, n, synth.ints_to_limbs([pow(n, -1, p) * R % p])[0])
        torch.cuda.synchronize()
        ok = (d.cpu().numpy().view(np.uint64) == a).all()
        # a'[0] = sum of the inputs (checked with the device inner product against the all-ones vector for sizes that fit)
        ones = torch.from_numpy(np.tile(synth.ints_to_limbs([R % p])[0], (n, 1)).view(np.int64)).cuda()
        s = api.inner_product_dev(field, torch.from_numpy(a.view(np.int64).copy()).cuda(), ones, n)
        ok0 = (s == fwd0).all()
        if not (ok and ok0):
            bad += 1
        print(f"ntt {field} 2^{log_n}: round trip {'ok' if ok else 'MISMATCH'}, a'[0] {'ok' if ok0 else 'MISMATCH'}", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
