"""Timeline of the single-call host entries (TRH_TRACE=1, read once per process, makes csrc/hostio.hip print microseconds since the call began):
tools/io_trace_probe.py [log_n]  -- a full 2^log_n best_fft, then the zero-padded shape of coeff_to_extended (data in the first eighth)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["TRH_TRACE"] = "1"
import torch
from tiny_ram_halo2_amd import api, synth
import pasta as o
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
api.init(0)
f = o.FIELDS["fp"]
w = np.array(f.limbs(f.omega(log_n)), np.uint64)
a = synth.ntt_input(log_n)
work = a.copy()
for rep in range(4):
    work[:] = a
    t0 = time.perf_counter(); api.best_fft_inplace("fp", work, w, log_n); dt = time.perf_counter() - t0
    print(f"full 2^{log_n}: {dt * 1e3:.3f} ms", file=sys.stderr)
pad = np.zeros_like(a)
for rep in range(4):
    pad[:] = 0
    pad[: a.shape[0] // 8] = a[: a.shape[0] // 8]
    api.io_stats(reset=True)
    t0 = time.perf_counter(); api.best_fft_inplace("fp", pad, w, log_n); dt = time.perf_counter() - t0
    print(f"zero-padded 2^{log_n} (1/8 data): {dt * 1e3:.3f} ms  {api.io_stats()}", file=sys.stderr)
full = np.zeros_like(a); full[: a.shape[0] // 8] = a[: a.shape[0] // 8]
d = torch.from_numpy(full.view(np.int64).copy()).cuda()
api.ntt_dev("fp", d, log_n, w); torch.cuda.synchronize()
assert (d.cpu().numpy().view(np.uint64) == pad).all(), "zero-elided upload changed the transform"
print("ok", file=sys.stderr)
# the shapes of the k = 18 proof's literal drop-in: trh_msm over 2^18 + 1 host scalars, best_fft 2^18, zero-padded best_fft 2^21 (wall times, no trace)
k = 18
n = 1 << k
bases = api.Bases.generate("vesta", synth.BASE_S0, synth.BASE_D, n + 1)
bases.precompute(0)
sc = synth.field_elements(0x18, n + 1)
w18 = np.array(f.limbs(f.omega(k)), np.uint64)
w21 = np.array(f.limbs(f.omega(k + 3)), np.uint64)
col = synth.field_elements(0x19, n)
ext = np.zeros((8 * n, 4), dtype=np.uint64)
def t(fn, reps=20):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3
def padded():
    ext[:n] = col; ext[n:] = 0
    t0 = time.perf_counter(); api.best_fft_inplace("fp", ext, w21, k + 3); return time.perf_counter() - t0
padded(); padded()
print(f"k=18 shapes: trh_msm 2^18+1 host scalars {t(lambda: bases.msm(sc)):.3f} ms; best_fft 2^18 {t(lambda: api.best_fft_inplace('fp', col, w18, k)):.3f} ms; "
      f"zero-padded best_fft 2^21 {sum(padded() for _ in range(20)) / 20 * 1e3:.3f} ms", file=sys.stderr)
print("--- trace: best_fft 2^18", file=sys.stderr)
api.best_fft_inplace("fp", col, w18, k)
print("--- trace: zero-padded best_fft 2^21", file=sys.stderr)
padded()
