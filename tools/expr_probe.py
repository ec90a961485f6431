"""Throughput of the gate evaluator at create_proof scale: tools/expr_probe.py [log_n] [n_advice] [n_gates]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_ram_halo2_amd import api, expr, synth
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n_adv = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_gates = int(sys.argv[3]) if len(sys.argv) > 3 else 150
api.init(0)
n = 1 << log_n
gates = expr.synthetic_gates(n_adv, 8, n_gates)
prog = expr.compile_gates("fp", gates, 12345)
ops = {k: int((prog.insns[:, 0] == v).sum()) for k, v in expr.OP.items()}
base = torch.from_numpy(synth.field_elements(0xE0, n).view(np.int64)).cuda()
cols = {key: torch.roll(base, i * 977 + 1, 0).contiguous() for i, key in enumerate(prog.columns)}
ev = expr.GateEvaluator(prog)
out = ev.eval(cols, log_n, 8)
torch.cuda.synchronize(); t = time.perf_counter()
reps = 3
for _ in range(reps):
    ev.eval(cols, log_n, 8, out=out)
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / reps * 1e3
muls = ops["MUL"] + ops["SQR"] + ops["MUL_CONST"] + ops["FOLD"]
loads = ops["PUSH_COLUMN"]
print(f"2^{log_n} rows, {len(prog.columns)} columns, {n_gates} gates, {len(prog.insns)} instructions ({muls} muls, {loads} column reads), LDS slots {ev.lds_slots()}: "
      f"{ms:.2f} ms  = {muls * n / ms / 1e6:.1f} G mul/s, {loads * n * 32 / ms / 1e6:.1f} GB/s of column reads")
