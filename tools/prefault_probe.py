"""Downloads into untouched pages: trh_best_fft on a zero-padded 2^21 vector (the literal coeff_to_extended of k = 18) and
trh_domain_coeff_to_extended_host into 2^21-element outputs, with the destination REUSED (its pages written before) and FRESH (a new calloc'ed
array per call: no page-table entries beyond what the caller stored -- what `vec![F::zero(); n]` of a Rust prover is).  Run twice:
    TRH_PREFAULT=0 python3 tools/prefault_probe.py;  TRH_PREFAULT=1 python3 tools/prefault_probe.py"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, poly, synth
api.init(0)
k, ek = 18, 21
n, N = 1 << k, 1 << ek
dom = poly.EvaluationDomain("fp", 6, k)
w_ext = dom._w["extended_omega"]
coeff = synth.field_elements(5, n)
out = {"TRH_PREFAULT": os.environ.get("TRH_PREFAULT", "1")}

def run(fresh, reps=12):
    reused = np.zeros((N, 4), dtype=np.uint64)
    reused[:] = 1  # every page written
    ts = []
    for r in range(reps + 2):
        a = np.zeros((N, 4), dtype=np.uint64) if fresh else reused
        a[:n] = coeff
        if not fresh:
            a[n:] = 0
        t0 = time.perf_counter()
        api.best_fft_inplace("fp", a, w_ext, ek)
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts[2:])) * 1e3, 3)

out["best_fft_2^21_padded_ms"] = {"reused": run(False), "fresh": run(True)}
cols = [coeff.copy() for _ in range(8)]
def run_ext(fresh, reps=4):
    bufs = [np.ones((N, 4), dtype=np.uint64) for _ in range(8)]
    ts = []
    for r in range(reps + 1):
        o = [np.zeros((N, 4), dtype=np.uint64) for _ in range(8)] if fresh else bufs
        t0 = time.perf_counter()
        dom.coeff_to_extended_host(cols, out=o)
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts[1:])) * 1e3 / 8, 3)
out["coeff_to_extended_host_ms_per_column"] = {"reused": run_ext(False), "fresh": run_ext(True)}
print(json.dumps(out))
