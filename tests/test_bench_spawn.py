"""CPU test (no GPU needed): `python bench.py --gpus N` without a launcher -- the shape of the command the driver runs -- must ALWAYS end
with one JSON line: on a box where the ranks cannot come up (here: no GPU at all) the parent, which never touches a GPU itself, reports
`"collective": {"ok": false, ...}` and exits non-zero instead of printing nothing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_without_a_launcher_reports_ranks_that_could_not_start():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this is the no-GPU behaviour; tests/test_gpu_native.py covers the GPU box")
    env = dict(os.environ, TRH_BENCH_SPAWN_GRACE="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] is None and out["collective"]["ok"] is False and out["collective"]["world"] == 2
    assert "exit codes" in out["collective"]["error"]
