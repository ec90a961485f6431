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
    step, the strong-scaling 2^26 leg and the single-process device-group leg all close their closed-form checks (exit code 0), and the
    proof-per-GPU leg reports its scaling figure"""
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
    # round 6: one k = 18 resident proof per rank (the reference's N independent proofs, /root/reference/src/test_utils.rs:37-54), MAX over ranks
    sc = out["e2e_summary"]["e2e_scaling"]
    assert sc["n_gpus"] == 2 and sc["ms_per_proof_slowest_rank"] >= sc["ms_this_rank"] > 0 and abs(sc["proofs_per_s"] - 2e3 / sc["ms_per_proof_slowest_rank"]) < 1e-6, sc
    assert out["e2e_summary"]["ntt_2_16_elems_per_s"] > 0 and 0 < out["e2e_summary"]["ntt_frac"] < 1


@pytest.mark.parametrize("args", [["--word-bits", "16", "--batch", "32", "--devices", "0,0"], ["--word-bits", "32", "--batch", "32", "--max-columns", "120", "--devices", "0,0,0"]])
def test_native_replay_column_sharded(args):
    """examples/replay --devices: the per-column phase column-sharded over one thread + trh::Context + Params copy per listed device from
    one compiled process; exit code 0 = the commitments and evaluations equal the single-context run's bit for bit"""
    exe = os.path.join(ROOT, "examples", "replay")
    r = subprocess.run([exe, "--columns", "witness", *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr + r.stdout
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["mode"] == "column-sharded" and out["commitments_identical_to_single_context"] is True
    n_dev = len(args[-1].split(","))
    assert len(out["per_device"]) == n_dev and out["per_device"][0]["columns"][0] == 0 and out["per_device"][-1]["columns"][1] == out["columns_replayed"]
    assert all(d["wall_ms"] > 0 for d in out["per_device"]) and out["single_context"]["wall_ms"] > 0
    print(out)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's command): the parent starts the two ranks itself before it
    touches a GPU, relays rank 0's line; the line names the backend that carried the partials, the world size it proved with an all-reduce
    and the devices.  gloo with the ranks folded onto the one GPU of the box; the driver's scaling runs use RCCL"""
    import sys
    env = dict(os.environ, TRH_BENCH_BACKEND="gloo", TRH_BENCH_SINGLE_PROCESS="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "18", "--ntt-log-n", "14", "--no-sweep"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    out = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["check"] == "closed-form ok" and out["config"]["pairs_total"] == 2 << 18
    c = out["collective"]
    assert c["ok"] is True and c["backend"] == "gloo" and c["world"] == 2 and c["all_reduce_of_ones"] == 2 and len(c["devices"]) == 2
    assert "gloo" in out["config"]["parallelism"] and "RCCL" not in out["config"]["parallelism"]


def test_bench_reports_a_collective_that_did_not_come_up():
    """the same command with the real backend on a box that has ONE GPU: rank 1 cannot bind its device, the collective never forms -- the
    run must say so in its line ("ok": false) and exit non-zero; it must not fall back to another backend or print a rate"""
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = dict(os.environ, TRH_BENCH_SPAWN_GRACE="5", TRH_BENCH_INIT_TIMEOUT="20")
    for k in ("WORLD_SIZE", "RANK", "TRH_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--log-n", "16", "--no-sweep"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, r.stdout + r.stderr[-2000:]
    out = json.loads(lines[-1])
    assert out["collective"]["ok"] is False and out["value"] is None and out["n_gpus"] == 2


def test_bench_one_rank_rccl_first_contact():
    """first-contact insurance for the driver's 8-GPU run: bench.py under a launcher's environment with ONE rank and the real
    backend -- init_process_group("nccl", device_id=...) = RCCL on this MI355X, the all-reduce-of-ones on a device tensor and
    sharded.all_gather_points on a device tensor -- everything the N > 1 path calls except a second device"""
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TRH_BENCH_BACKEND="nccl",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--log-n", "16", "--no-sweep", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    out = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    c = out["collective"]
    assert c["backend"] == "nccl" and c["ok"] is True and c["world"] == 1 and c["all_reduce_of_ones"] == 1, c
    assert c["all_gather_points"] == "ok" and out["check"] == "closed-form ok", (c, out["check"])
