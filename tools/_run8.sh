python -m pytest tests/test_gpu_dropin.py -x -q 2>&1 | tail -5
python tools/io_trace_probe.py 22 2>&1 | grep -E "^full|^zero-padded|k=18 shapes|^---|before-copy|speculat|upload issued|transform|download complete|end" | tail -40
python -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --mode dropin 2>/dev/null | tail -1 > gpurun_out/replay_dropin_r04b.json
python -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --mode dropin-batched 2>/dev/null | tail -1 > gpurun_out/replay_dropin-batched_r04b.json
python - <<'PY'
import json
for f in ("gpurun_out/replay_dropin_r04b.json","gpurun_out/replay_dropin-batched_r04b.json"):
    d=json.load(open(f)); print(f, d["wall_ms_incl_pcie_total"], d["wall_ms_incl_pcie"], d["pcie"])
PY
