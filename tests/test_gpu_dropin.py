"""GPU tests (-m gpu) of the host-pointer ("drop-in") side of the C ABI -- csrc/hostio.hip: pinned staging rings, the tiled
host-scalar / host-bases MSM, the pipelined batch entries -- and of the drop-in create_proof replay at k = 10, every call
compared with the oracle as tests/test_gpu_replay.py does for the resident replay.
Reference seam: halo2_proofs::arithmetic::{best_multiexp, best_fft}, Params::commit_lagrange, EvaluationDomain::* as
create_proof calls them with host slices (/root/reference/src/test_utils.rs:41-49)."""
import ctypes
import os

import numpy as np
import pytest

import cpu_ref
import pasta as o
from tiny_ram_halo2_amd import api, poly, replay, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init(0)
    yield


def _affine(curve, jac):
    return np.asarray(jac, dtype=np.uint64)[:8]


@pytest.mark.parametrize("level", ["literal", "batched", "batched-blocks"])
def test_dropin_replay_k10_matches_oracle(level):
    """BASELINE config 1's circuit size through the host-pointer entries: every hooked call against the oracle"""
    seen = {}

    def hook(kind, inp, out):
        seen[kind] = seen.get(kind, 0) + 1
        if kind in ("commit_lagrange", "commit"):
            want = cpu_ref.to_affine("vesta", cpu_ref.best_multiexp("vesta", inp["scalars"], inp["bases"].download(), threads=8))
            assert (np.asarray(out)[:8] == want).all(), kind
            return
        if kind == "best_multiexp":
            want = cpu_ref.to_affine(inp["curve"], cpu_ref.best_multiexp(inp["curve"], inp["scalars"], inp["bases"], threads=8))
            assert (np.asarray(out)[:8] == want).all(), kind
            return
        if kind == "best_fft":
            assert (np.asarray(out) == cpu_ref.best_fft(inp["field"], inp["a"], inp["omega"], inp["log_n"], threads=8)).all()
            return
        if kind == "best_fft_padded":  # the zero-padded 2^extended_k vector of coeff_to_extended
            full = np.zeros((1 << inp["log_n"], 4), dtype=np.uint64)
            full[: inp["a"].shape[0]] = inp["a"]
            assert (np.asarray(out) == cpu_ref.best_fft(inp["field"], full, inp["omega"], inp["log_n"], threads=8)).all()
            return
        if kind in ("coeff_to_extended_blocks", "blocks_to_quotient"):
            from test_gpu_replay import check_blocks
            check_blocks(kind, inp, out)
            return
        field, j, k = inp["domain"]
        f = o.FIELDS[field]
        dom = o.EvaluationDomain(f, j, k)
        a = [f.from_limbs(r) for r in np.asarray(inp["a"]).reshape(-1, 4)]
        if kind == "lagrange_to_coeff":
            want = dom.lagrange_to_coeff(a)
        elif kind == "coeff_to_extended":
            want = dom.coeff_to_extended(a)
        else:
            want = dom.extended_to_coeff(dom.divide_by_vanishing_poly(a))
        assert [f.from_limbs(r) for r in np.asarray(out).reshape(-1, 4)] == want, kind

    res = replay.run_dropin(16, level, batch=32, hook=hook, verbose=False, columns="witness")
    assert res["schedule"]["k"] == 10 and res["counts"]["commit_lagrange"] == 497 and res["counts"]["coeff_to_extended"] == 497
    assert res["pcie"]["h2d_GB"] > 0 and res["pcie"]["d2h_GB"] > 0 and res["wall_ms_incl_pcie_total"] > 0
    if level == "literal":
        assert seen == {"commit_lagrange": 3, "best_fft": 4, "best_fft_padded": 1, "commit": 1, "best_multiexp": 4}
        # the literal level moves every padded vector both ways: 497 x (n + 1 + n + 8 n) x 32 B up
        n = 1 << 10
        assert res["pcie"]["h2d_GB"] * 1e9 >= 497 * (10 * n + 1) * 32
    elif level == "batched":
        assert seen == {"commit_lagrange": 3, "lagrange_to_coeff": 3, "coeff_to_extended": 3, "commit": 1, "divide_and_extended_to_coeff": 1}
    else:
        assert seen == {"commit_lagrange": 3, "lagrange_to_coeff": 3, "coeff_to_extended_blocks": 3, "commit": 1, "blocks_to_quotient": 1}
        assert res["mode"] == "dropin-batched-blocks"


@pytest.mark.parametrize("curve", ["pallas", "vesta"])
def test_host_msm_tiles_vs_oracle(curve, monkeypatch):
    """trh_best_multiexp_* (host scalars AND host bases) and trh_msm (resident bases, host scalars) cut into ranges -- forced small
    here so that several tiles, a ragged last one and the double-buffered uploads are exercised -- against cpu_ref.best_multiexp
    with unstructured bases"""
    n = (1 << 16) + 13
    sc = synth.field_elements(0x51, n)
    bases = cpu_ref.gen_bases_hashed(curve, 11, n)
    want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=8))
    res = api.Bases.from_host(curve, bases)
    for tile_log in (None, 14, 12):
        if tile_log is None:
            monkeypatch.delenv("TRH_HOST_TILE_LOG", raising=False)
        else:
            monkeypatch.setenv("TRH_HOST_TILE_LOG", str(tile_log))
        assert (api.best_multiexp(curve, sc, bases)[:8] == want).all(), ("best_multiexp", tile_log)
        assert (res.msm(sc)[:8] == want).all(), ("msm", tile_log)
        off, ln = 1000, (1 << 15) + 7  # a sub-range of the resident set, tiled
        w2 = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc[:ln], bases[off:off + ln], threads=8))
        assert (res.msm(sc[:ln], offset=off)[:8] == w2).all(), ("msm offset", tile_log)
    monkeypatch.delenv("TRH_HOST_TILE_LOG", raising=False)
    # degenerate sizes through the same path
    ident = np.zeros(8, dtype=np.uint64)
    assert (api.best_multiexp(curve, sc[:0], bases[:0])[:8] == ident).all()
    assert (api.best_multiexp(curve, sc[:1], bases[:1])[:8] == cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc[:1], bases[:1], threads=1))).all()


@pytest.mark.parametrize("field,log_n,count", [("fp", 12, 37), ("fq", 16, 9), ("fp", 20, 5)])
def test_best_fft_batch_host_vs_oracle(field, log_n, count):
    """the pipelined batch form == `count` single transforms == the oracle; small columns travel in groups, large ones alone"""
    f = o.FIELDS[field]
    w = np.array(f.limbs(f.omega(log_n)), np.uint64)
    cols = [np.ascontiguousarray(synth.field_elements(0xF00 + i, 1 << log_n)) for i in range(count)]
    want = [cpu_ref.best_fft(field, c, w, log_n, threads=8) for c in cols[:3]] + [None] * (count - 3)
    single = [api.best_fft(field, c, w, log_n) for c in cols]
    api.best_fft_batch(field, cols, w, log_n)
    for i in range(count):
        assert (cols[i] == single[i]).all(), i
        if want[i] is not None:
            assert (cols[i] == want[i]).all(), i


def test_commit_batch_host_vs_single_commits():
    """trh_commit_batch_host (chunked uploads under the batched MSMs, blinds riding behind each chunk) == Params::commit_lagrange
    column by column == the oracle, with and without fixed-base tables, batch sizes around the chunk size"""
    k, n = 12, 1 << 12
    curve = "vesta"
    bases = cpu_ref.gen_bases_hashed(curve, 5, n + 1)
    b = api.Bases.from_host(curve, bases)
    for tables in (False, True):
        if tables:
            b.precompute(0)
        for batch in (1, 15, 16, 17, 40):
            cols = [np.ascontiguousarray(synth.field_elements(0xA0 + i, n)) for i in range(batch)]
            blinds = synth.field_elements(0xB0, batch)
            got = b.commit_batch_host(cols, blinds)
            for i in (0, batch // 2, batch - 1):
                sc = np.concatenate([cols[i], blinds[i][None]])
                assert (got[i] == b.msm(sc)).all(), (tables, batch, i)
            want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, np.concatenate([cols[0], blinds[0][None]]), bases, threads=8))
            assert (got[0][:8] == want).all()
    del k


def test_domain_host_forms_vs_oracle():
    """EvaluationDomain on host polynomials: lagrange_to_coeff / coeff_to_extended batches and h(X)'s divide + extended_to_coeff"""
    field, j, k = "fp", 6, 11
    dom = poly.EvaluationDomain(field, j, k)
    cd = cpu_ref.EvaluationDomain(field, j, k)
    n, N = 1 << k, 1 << dom.extended_k
    cols = [np.ascontiguousarray(synth.field_elements(0xD0 + i, n)) for i in range(11)]
    orig = [c.copy() for c in cols]
    dom.lagrange_to_coeff_host(cols)
    for i in (0, 5, 10):
        assert (cols[i] == cd.lagrange_to_coeff(orig[i])).all(), i
    ext = dom.coeff_to_extended_host(cols)
    for i in (0, 10):
        assert ext[i].shape == (N, 4) and (ext[i] == cd.coeff_to_extended(cols[i])).all(), i
    h = np.ascontiguousarray(synth.field_elements(0xEE, N))
    want = cd.extended_to_coeff(cd.divide_by_vanishing_poly(h.copy()))
    got = dom.extended_to_coeff_host(h, divide_by_vanishing_first=True)
    assert (got == np.asarray(want).reshape(-1, 4)[: got.shape[0]]).all()


def test_pinned_caller_memory_skips_the_ring():
    """memory from trh_host_alloc / trh_host_register is read and written by the DMA engine directly: same results"""
    field, log_n = "fp", 16
    f = o.FIELDS[field]
    w = np.array(f.limbs(f.omega(log_n)), np.uint64)
    src = synth.field_elements(0x9191, 1 << log_n)
    want = api.best_fft(field, src, w, log_n)
    p = ctypes.c_void_p()
    api._check(api.lib().trh_host_alloc(ctypes.byref(p), src.nbytes))
    try:
        a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint64)), shape=(1 << log_n, 4))
        a[:] = src
        api.io_stats(reset=True)
        api.best_fft_inplace(field, a, w, log_n)
        assert (a == want).all()
        st = api.io_stats()
        assert st["h2d_bytes"] == src.nbytes and st["d2h_bytes"] == src.nbytes
    finally:
        api._check(api.lib().trh_host_free(p))
    reg = np.ascontiguousarray(src.copy())
    api._check(api.lib().trh_host_register(reg.ctypes.data_as(ctypes.c_void_p), reg.nbytes))
    try:
        api.best_fft_inplace(field, reg, w, log_n)
        assert (reg == want).all()
    finally:
        api._check(api.lib().trh_host_unregister(reg.ctypes.data_as(ctypes.c_void_p)))


def test_busy_context_is_reported_and_recovers():
    """trh_msm on a context with an enqueued MSM answers TRH_EBUSY (it used to overwrite it); afterwards the context works"""
    import torch
    n = 1 << 12
    b = api.Bases.generate("pallas", synth.BASE_S0, synth.BASE_D, n)
    sc = synth.msm_scalars(12)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    b.msm_dev_enqueue(d_sc, n)
    with pytest.raises(api.TrhError):
        b.msm(sc)
    got = b.msm_dev_finish()
    assert (b.msm(sc) == got).all()
    os.environ.pop("TRH_HOST_TILE_LOG", None)


def test_device_block_pool_reuses_and_isolates():
    """trh_malloc / trh_free keep freed blocks per (device, rounded size): the same block comes back, live blocks are never shared,
    contents written through one block are read back from it"""
    lib = api.lib()
    size = (3 << 20) + 17
    p, q, r = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    api._check(lib.trh_malloc(ctypes.byref(p), size))
    api._check(lib.trh_malloc(ctypes.byref(q), size))
    assert p.value and q.value and p.value != q.value
    src = np.arange(size // 8, dtype=np.uint64)
    api._check(lib.trh_memcpy_h2d(p, src.ctypes.data_as(ctypes.c_void_p), src.nbytes))
    first = p.value
    api._check(lib.trh_free(p))
    api._check(lib.trh_malloc(ctypes.byref(r), size - 5))   # same rounded size: the idle block
    if os.environ.get("TRH_POOL_MB", "1") != "0":
        assert r.value == first
    assert r.value != q.value
    back = np.zeros_like(src)
    fits = (size - 5) // 8 * 8   # r was asked for size - 5 bytes: without the pool that is all it has
    api._check(lib.trh_memcpy_h2d(r, src.ctypes.data_as(ctypes.c_void_p), fits))
    api._check(lib.trh_memcpy_d2h(back.ctypes.data_as(ctypes.c_void_p), r, fits))
    assert (back[: fits // 8] == src[: fits // 8]).all()
    api._check(lib.trh_free(r))
    api._check(lib.trh_free(q))
    assert lib.trh_free(None) == 0


def test_best_fft_zero_padded_vectors_and_failed_speculation():
    """trh_best_fft on the zero-padded vector coeff_to_extended hands over (data in the first eighth): the padding is not sent -- chunks
    that probe as zero are cleared on the device at once and read through while the device works (csrc/hostio.hip best_fft_host).
    Against the oracle's best_fft: the plain padded vector; a vector with ONE non-zero element deep inside the padding, off every probed
    line (the speculation must fail and the call start over); an all-zero vector; a full vector"""
    field, log_n = "fp", 20
    n = 1 << log_n
    f = o.FIELDS[field]
    w = np.array(f.limbs(f.omega(log_n)), np.uint64)
    data = synth.field_elements(0x2E20, n // 8)
    th = cpu_ref.hardware_threads()
    padded = np.zeros((n, 4), dtype=np.uint64)
    padded[: n // 8] = data
    hidden = padded.copy()
    hidden[5 * n // 8 + 3] = data[7]          # byte offset 20 MiB + 96: not a multiple of 64 KiB, not a chunk's first or last line
    edge = padded.copy()
    edge[n - 1] = data[9]                     # the very last element
    for vec in (padded, hidden, edge, np.zeros((n, 4), dtype=np.uint64), synth.field_elements(0x2E21, n)):
        want = cpu_ref.best_fft(field, vec, w, log_n, threads=th)
        work = vec.copy()
        api.io_stats(reset=True)
        api.best_fft_inplace(field, work, w, log_n)
        assert (work == want).all()
    io = api.io_stats()
    assert io["h2d_zero_bytes"] == 0          # the last vector is full: nothing elided
    work = padded.copy()
    api.io_stats(reset=True)
    api.best_fft_inplace(field, work, w, log_n)
    io = api.io_stats()
    assert io["h2d_zero_bytes"] >= (n - n // 8 - (1 << 19)) * 32 and io["h2d_bytes"] == n * 32, io
