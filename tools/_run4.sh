python -m pytest tests/test_gpu_sparse.py -x -q 2>&1 | tail -15
for nt in 0 1 3; do for th in 4 8; do echo "NT=$nt THREADS=$th"; TRH_NT_COPY=$nt TRH_COPY_THREADS=$th python tools/io_trace_probe.py 22 2>&1 | grep -E "^full|^zero-padded|k=18 shapes" | tail -4; done; done
python -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --no-keygen 2>/dev/null | tail -1 > gpurun_out/replay_witness_r04a.json
TRH_SPARSE=0 python -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --no-keygen 2>/dev/null | tail -1 > gpurun_out/replay_witness_r04a_nosparse.json
python - <<'PY'
import json
for f in ("gpurun_out/replay_witness_r04a.json","gpurun_out/replay_witness_r04a_nosparse.json"):
    d=json.load(open(f)); print(f, d["gpu_ms_total"], d["gpu_ms"])
PY
python -m pytest tests/test_gpu_replay.py tests/test_gpu_ipa.py -x -q 2>&1 | tail -5
