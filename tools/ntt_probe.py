"""Small driver for profiling the NTT alone: tools/ntt_probe.py [log_n] [reps] [batch]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, synth
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
api.init(0)
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
root = 0x2BCE74DEAC30EBDA362120830561F81AEA322BF2B7BB7584BDAD6FABD87EA32F
w = synth.ints_to_limbs([pow(root, 1 << (32 - log_n), P) * ((1 << 256) % P) % P])[0]
a = synth.field_elements(7, (1 << log_n) * batch)
d = torch.from_numpy(a.view(np.int64).copy()).cuda()
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    api.ntt_dev("fp", d, log_n, w, batch=batch, stream=st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(reps):
    api.ntt_dev("fp", d, log_n, w, batch=batch, stream=st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
import hashlib
d2 = torch.from_numpy(a.view(np.int64).copy()).cuda()
api.ntt_dev("fp", d2, log_n, w, batch=batch, stream=st)
torch.cuda.synchronize()
digest = hashlib.sha256(d2.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"ntt 2^{log_n} x{batch}: {ms:.4f} ms  {(1 << log_n) * batch / ms / 1e6:.2f} Gelem/s  output sha256 {digest}")
