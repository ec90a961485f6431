"""Host-pointer MSM / FFT entry points at the headline sizes: wall time of trh_best_multiexp_pallas (host scalars AND
host bases), trh_msm (resident bases, host scalars) and trh_best_fft_fp against the two things they are bounded by -- the link (bytes /
57 GB/s, the pinned-copy rate of the box) and the resident kernel time.   tools/dropin_probe.py [log_n_msm] [log_n_fft]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from tiny_ram_halo2_amd import api, synth
import pasta as o

LINK = 57.0e9
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
log_f = int(sys.argv[2]) if len(sys.argv) > 2 else 22
api.init(0)
n = 1 << log_n
curve = "pallas"
res = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
bases = res.download()          # (n, 8) host, pageable
sc = synth.msm_scalars(log_n)   # (n, 4) host
d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
st = torch.cuda.current_stream().cuda_stream
out = {"log_n": log_n}

def timeit(f, reps=5, warm=2):
    for _ in range(warm):
        r = f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return r, min(ts) * 1e3, sum(ts) / len(ts) * 1e3

ref, res_min, res_avg = timeit(lambda: res.msm_dev(d_sc, n, stream=st))
out["resident_msm_ms"] = round(res_min, 3)
api.io_stats(reset=True)
r1, bm_min, bm_avg = timeit(lambda: api.best_multiexp(curve, sc, bases))
io = api.io_stats(reset=True)
assert (r1 == ref).all(), "best_multiexp (host) != resident MSM"
link_ms = n * 96 / LINK * 1e3
out["best_multiexp_host"] = {"ms_min": round(bm_min, 3), "ms_avg": round(bm_avg, 3), "link_ms_at_57GBps": round(link_ms, 3), "bound_ms": round(max(link_ms, res_min), 3),
                             "over_bound": round(bm_min / max(link_ms, res_min), 3), "sum_ms": round(link_ms + res_min, 3),
                             "h2d_GBps_in_copies": round(io["h2d_bytes"] / io["h2d_seconds"] / 1e9, 2), "GBps_over_call": round(n * 96 / (bm_min * 1e-3) / 1e9, 2)}
r2, m_min, m_avg = timeit(lambda: res.msm(sc))
assert (r2 == ref).all(), "trh_msm (host scalars) != resident MSM"
link2 = n * 32 / LINK * 1e3
out["msm_host_scalars"] = {"ms_min": round(m_min, 3), "ms_avg": round(m_avg, 3), "link_ms_at_57GBps": round(link2, 3), "bound_ms": round(max(link2, res_min), 3),
                           "over_bound": round(m_min / max(link2, res_min), 3), "sum_ms": round(link2 + res_min, 3)}
del bases, sc, d_sc, res
# closed-form check of the reference point is bench.py's business; here: FFT
f = o.FIELDS["fp"]
w = np.array(f.limbs(f.omega(log_f)), np.uint64)
a = synth.ntt_input(log_f)
d_a = torch.from_numpy(a.view(np.int64).copy()).cuda()
def ntt_res():
    api.ntt_dev("fp", d_a, log_f, w, stream=st); torch.cuda.synchronize()
_, k_min, _ = timeit(ntt_res)
work = a.copy()
def fft_host():
    work[:] = a
    t0 = time.perf_counter(); api.best_fft_inplace("fp", work, w, log_f); return time.perf_counter() - t0
ts = [fft_host() for _ in range(6)][1:]
d_b = torch.from_numpy(a.view(np.int64).copy()).cuda()
api.ntt_dev("fp", d_b, log_f, w, stream=st); torch.cuda.synchronize()
assert (work == d_b.cpu().numpy().view(np.uint64)).all(), "best_fft (host) != resident NTT"
bytes_each = (32 << log_f)
out["best_fft_host"] = {"log_n": log_f, "ms_min": round(min(ts) * 1e3, 3), "resident_ntt_ms": round(k_min, 3), "link_ms_each_way": round(bytes_each / LINK * 1e3, 3),
                        "sum_ms": round(2 * bytes_each / LINK * 1e3 + k_min, 3)}
# batch of 16 columns, pipelined both ways
cols = [a.copy() for _ in range(16)]
t0 = time.perf_counter(); api.best_fft_batch("fp", cols, w, log_f); dt = time.perf_counter() - t0
cols = [a.copy() for _ in range(16)]
t0 = time.perf_counter(); api.best_fft_batch("fp", cols, w, log_f); dt = min(dt, time.perf_counter() - t0)
assert (cols[7] == work).all()
out["best_fft_batch_host"] = {"log_n": log_f, "columns": 16, "ms_per_column": round(dt * 1e3 / 16, 3), "GBps_each_way": round(16 * bytes_each / dt / 1e9, 2)}
print(json.dumps(out))
