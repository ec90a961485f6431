"""GPU test (-m gpu): the native C++ driver (examples/replay.cpp over include/trh.hpp) -- the host side of the boundary as a
compiled caller, no Python in the process.  It runs the create_proof schedule at k = 10 and checks one item of every
primitive kind itself (host-side field arithmetic of trh.hpp / a second libtrh path); a non-zero exit code is a failed check."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("columns", ["random", "witness"])
def test_native_replay_k10(columns):
    exe = os.path.join(ROOT, "examples", "replay")
    if not os.path.exists(exe):  # normally built by `make` / __graft_entry__.build(); g++ only, libtrh.so must already be there
        subprocess.check_call(["make", "-s", "-C", ROOT, "examples/replay"])
    r = subprocess.run([exe, "--word-bits", "16", "--batch", "32", "--columns", columns], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["k"] == 10 and out["checks_failed"] == 0 and out["columns"] == columns and out["keygen_ms"] > 0
    assert set(out["ms"]) == {"lookup_permute", "commit_lagrange", "lagrange_to_coeff", "coeff_to_extended", "evals", "h_eval", "commit", "extended_to_coeff", "multiopen_folds", "ipa"}


def test_native_multi_context():
    """tests/native/multi_ctx_test.cpp: device group ({0, 0} on a one-GPU box) with range-sharded base sets through trh_msm /
    trh_msm_dev / trh_best_multiexp_*, two host threads on two contexts overlapping MSMs and NTTs, enqueue / finish bookkeeping,
    a non-init thread on the last device -- from a compiled host, no Python in the process"""
    exe = os.path.join(ROOT, "tests", "native", "multi_ctx_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "tests/native/multi_ctx_test"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["checks_failed"] == 0 and out["group"] >= 2


def test_bench_two_ranks_on_one_device():
    """the process-per-GPU harness (bench.py under torch.distributed.run, sharded.sharded_msm over the HIP path) with two ranks
    folded onto the GPUs present (gloo carries the 96-byte partials; the driver's scaling runs use RCCL): headline weak-scaling
    step, the strong-scaling 2^26 leg and the single-process device-group leg all close their closed-form checks (exit code 0)"""
    import sys
    env = dict(os.environ, TRH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "20", "--ntt-log-n", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    out = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["check"] == "closed-form ok"
    assert out["config"]["pairs_total"] == 2 << 20
    assert out["strong"]["check"] == "closed-form ok" and out["strong"]["scaling"] == "strong"
    assert out["single_process"].get("check") == "closed-form ok", out["single_process"]
