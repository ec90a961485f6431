"""GPU tests (-m gpu) of the multi-GPU paths of the C ABI on the ONE GPU of the test box: a device group that lists device 0 several
times makes every shard its own context -- own stream, own pinned rings, own copy threads -- so the code that feeds and collects G
devices runs for real, only the links are shared (the reference proves in one process, /root/reference/src/test_utils.rs:37-54).
  * host scalars into a range-sharded base set: one uploader thread per shard (csrc/capi.hip msm_sharded), same point as one context;
  * option force_no_peer = 1 (a subprocess): device-resident scalars reach every shard through pinned host memory (stage_d2d_via_host), same point;
  * the block pool behind trh_malloc / trh_free with a tiny cap (TRH_POOL_MB=1, a subprocess): eviction, double free, trim.
The group is [0] * 8 everywhere in the suite (trh_init_multi accepts only the list it was first given)."""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

import cpu_ref
from tiny_ram_halo2_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = 8


@pytest.fixture()
def group():
    api.init_multi([0] * G)
    assert api.group_size() == G
    api.set_shard_min(1 << 12)
    yield
    api.set_shard_min(1 << 62)  # later tests of the session create single-device sets again
    torch.cuda.synchronize()


def _single(curve, n, sc):
    """the same MSM on one context (the set is created below the shard threshold)"""
    api.set_shard_min(1 << 62)
    b = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    assert b.shards() == 1
    want = b.msm(sc)
    b.destroy()
    api.set_shard_min(1 << 12)
    return want


@pytest.mark.parametrize("n", [(1 << 16) + 5, 1 << 22])
def test_sharded_host_scalars_match_one_context(group, n):
    """trh_msm with HOST scalars over a set range-sharded 8 ways: the ranges are uploaded by one thread per shard through that
    shard's own ring; the point is the single-context point (and the oracle's at the smaller size); a sub-range that touches only
    some shards, with an offset, agrees too"""
    curve = "pallas"
    sc = synth.field_elements(0x5A4D + n, n)
    want = _single(curve, n, sc)
    b = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    assert b.shards() == G
    assert (b.msm(sc) == want).all()
    if n <= (1 << 17):
        ref = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc, b.download(), threads=cpu_ref.hardware_threads()))
        assert (np.asarray(want)[:8] == ref).all()
    # page-locked caller memory goes to the DMA engines directly (no ring, no copy threads): same point
    pinned = torch.from_numpy(sc.view(np.int64)).pin_memory()
    assert (b.msm(pinned.numpy().view(np.uint64)) == want).all()
    lo, cnt = n // 3, n // 2
    part = b.msm(np.ascontiguousarray(sc[lo:lo + cnt]), offset=lo)
    api.set_shard_min(1 << 62)
    one = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    assert (one.msm(np.ascontiguousarray(sc[lo:lo + cnt]), offset=lo) == part).all()
    one.destroy()
    b.destroy()


def test_sharded_host_upload_is_not_serialised(group):
    """the host-scalar sharded call must not cost G x the single-shard call (the uploads used to run one after
    another from the calling thread through one process-wide copy pool).  On one GPU the shards share the link and the chip, so the
    bound checked is the loose one: the sharded call is no slower than G single-shard calls back to back"""
    curve, n = "pallas", 1 << 23
    sc = synth.field_elements(0x5A4E, n)
    b = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    assert b.shards() == G
    per = n // G
    api.set_shard_min(1 << 62)
    one = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, per)
    api.set_shard_min(1 << 12)
    first = np.ascontiguousarray(sc[:per])
    for _ in range(2):
        b.msm(sc), one.msm(first)
    t = time.perf_counter()
    for _ in range(3):
        b.msm(sc)
    sharded = (time.perf_counter() - t) / 3
    t = time.perf_counter()
    for _ in range(3):
        one.msm(first)
    single = (time.perf_counter() - t) / 3
    print(f"sharded host-scalar MSM 2^23 over {G} shards: {sharded * 1e3:.2f} ms; one shard's 2^20: {single * 1e3:.2f} ms")
    assert sharded <= G * single * 1.10, (sharded, single)
    one.destroy()
    b.destroy()


NO_PEER_SCRIPT = r"""
import numpy as np, torch
from tiny_ram_halo2_amd import api, synth
from common import point_hex
G = 8
api.init_multi([0] * G)
assert api.get_option("force_no_peer") == 1 and api.lib().trh_group_peer_access() == 0
api.set_shard_min(1 << 12)
curve, n = "pallas", (1 << 21) + 77
d = torch.from_numpy(synth.field_elements(0x5A4F, n).view(np.int64)).cuda()
b = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
assert b.shards() == G
api.io_stats(reset=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    d2 = d.clone()
    got = b.msm_dev(d2, n, stream=s.cuda_stream)
    again = b.msm_dev(d2, n, stream=s.cuda_stream)   # the ring is reused straight away
assert (got == again).all()
io = api.io_stats()   # shard 0 is the default context: its share went through its ring twice
assert io["h2d_bytes"] >= 2 * (n // G) * 32 and io["d2h_bytes"] >= 2 * (n // G) * 32, io
print("full", point_hex(got))
lo, cnt = n // 5, n // 2
print("part", point_hex(b.msm_dev(d[lo:lo + cnt].contiguous(), cnt, offset=lo)))
"""


def test_forced_no_peer_hand_over(group):
    """option force_no_peer = 1 (a fresh process: options are fixed while a context exists): device-resident scalars are handed to EVERY
    shard through the destination context's pinned ring (D2H on the caller's stream, H2D on the shard's, chained by events) -- the path
    a box without peer access takes; the points equal the peer / same-device path's of this process and the ring's byte counters moved"""
    from common import point_hex, run_with_options
    curve, n = "pallas", (1 << 21) + 77
    d = torch.from_numpy(synth.field_elements(0x5A4F, n).view(np.int64)).cuda()
    b = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    assert b.shards() == G
    want = b.msm_dev(d, n)
    lo, cnt = n // 5, n // 2
    want_part = b.msm_dev(d[lo:lo + cnt].contiguous(), cnt, offset=lo)
    b.destroy()
    out = dict(line.split() for line in run_with_options(NO_PEER_SCRIPT, {"TRH_FORCE_NO_PEER": "1"}).splitlines() if line.startswith(("full", "part")))
    assert out["full"] == point_hex(want) and out["part"] == point_hex(want_part)


POOL_SCRIPT = r"""
import ctypes, sys
sys.path.insert(0, %r)
from tiny_ram_halo2_amd import api
api.init(0)
lib = api.lib()
def alloc(n):
    p = ctypes.c_void_p()
    api._check(lib.trh_malloc(ctypes.byref(p), n))
    return p
MB = 1 << 20
assert lib.trh_pool_idle_bytes() == 0
a, b, c = alloc(MB // 2), alloc(MB // 4), alloc(3 * MB)
api._check(lib.trh_free(a))
assert lib.trh_pool_idle_bytes() == MB // 2                      # kept
assert lib.trh_free(a) != 0 and b"already freed" in lib.trh_last_error()   # double free: reported, the block stays pooled
assert lib.trh_pool_idle_bytes() == MB // 2
api._check(lib.trh_free(c))                                      # larger than the 1 MiB cap: goes back to the runtime
assert lib.trh_pool_idle_bytes() == MB // 2
d = alloc(3 * MB // 4)
api._check(lib.trh_free(d))                                      # 0.5 + 0.75 > 1 MiB: the largest idle block (0.5) makes room
assert lib.trh_pool_idle_bytes() == 3 * MB // 4, lib.trh_pool_idle_bytes()
api._check(lib.trh_free(b))                                      # 0.75 + 0.25 fits
assert lib.trh_pool_idle_bytes() == MB
e = alloc(MB // 4)
assert e.value == b.value                                        # the idle block of that class comes back
api._check(lib.trh_pool_trim())
assert lib.trh_pool_idle_bytes() == 0
api._check(lib.trh_free(e))
print("pool ok")
"""


def test_block_pool_with_a_tiny_cap():
    """TRH_POOL_MB=1 in a fresh process: blocks above the cap are not kept, making room evicts the largest idle block, a double free
    is reported and harmless, trh_pool_trim empties the pool (the two-device eviction order is csrc/devpool.h's CPU test)"""
    env = dict(os.environ, TRH_POOL_MB="1")
    r = subprocess.run([sys.executable, "-c", POOL_SCRIPT % ROOT], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and "pool ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("word_bits,devices,max_columns", [(16, [0, 0], None), (32, [0, 0, 0], 150)])
def test_replay_column_sharded_over_contexts(word_bits, devices, max_columns):
    """replay --devices 0,0: the per-column phase of the create_proof schedule column-sharded over one thread + context + Params copy
    per listed device.  The commitments and evaluations gathered in column order are the single-context run's, bit for
    bit; every device reports its own times.  k = 10 with all 497 columns over two contexts, k = 18 with 150 columns (instance flags /
    words and blinded advice flags) over three"""
    from tiny_ram_halo2_amd import replay
    r = replay.run_sharded(word_bits, devices, batch=32, columns="witness", verbose=False, max_columns=max_columns)
    assert r["commitments_identical_to_single_context"], r
    total = 497 if max_columns is None else max_columns
    assert r["columns_replayed"] == total and len(r["per_device"]) == len(devices)
    assert [d["columns"][0] for d in r["per_device"]][0] == 0 and r["per_device"][-1]["columns"][1] == total
    for a, b in zip(r["per_device"], r["per_device"][1:]):
        assert a["columns"][1] == b["columns"][0]           # contiguous ranges in order
    for d in r["per_device"]:
        assert d["wall_ms"] > 0 and all(v > 0 for v in d["gpu_ms"].values())
    print({k: r[k] for k in ("wall_ms", "single_context", "per_device")})
