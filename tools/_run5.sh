python -m pytest tests/test_gpu_sparse.py -x -q 2>&1 | tail -8
python -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --no-keygen 2>/dev/null | tail -1 > gpurun_out/replay_witness_r04b.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/replay_witness_r04b.json")); print(d["gpu_ms_total"], d["gpu_ms"])
PY
bash tools/prof_cmd.sh sparse_r04b tools/replay_probe.py 32 witness 2>&1 | head -30
python -m pytest tests/test_gpu_replay.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -5
