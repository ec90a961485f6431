"""GPU parity tests (-m gpu): the HIP path, called through the C ABI of libtrh.so, against the
oracle (oracle/pasta.py big-int, oracle/cpu_ref.cpp restatement of best_multiexp / best_fft) and
the committed golden vectors.  Bit-exact: every comparison is limb-for-limb equality.
Nothing here reads /root/reference.
"""
import numpy as np
import pytest

import cpu_ref
import pasta as o
from common import load_json, load_npz, unhex, unhex_rows
from tiny_ram_halo2_amd import api, synth

pytestmark = pytest.mark.gpu

FIELDS = ["fp", "fq"]
CURVES = ["pallas", "vesta"]


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init(0)
    yield


def aff(curve, jac):
    return cpu_ref.to_affine(curve, jac)


# ---------------------------------------------------------------------------------------
# K1 / K2: field and group arithmetic on the device
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("field", FIELDS)
def test_field_ops_golden(field):
    kat = load_json("field_kat.json")[field]
    a = unhex_rows([r["a"] for r in kat["rows"]])
    b = unhex_rows([r["b"] for r in kat["rows"]])
    for op in ("add", "sub", "mul"):
        assert (api.field_op_dev(field, op, a, b) == unhex_rows([r[op] for r in kat["rows"]])).all(), op
    for op in ("sqr", "neg", "inv"):
        assert (api.field_op_dev(field, op, a) == unhex_rows([r[op] for r in kat["rows"]])).all(), op
    can = api.field_op_dev(field, "from_mont", a)
    assert (can == unhex_rows([r["a_canonical"] for r in kat["rows"]])).all()
    assert (api.field_op_dev(field, "to_mont", can) == a).all()


@pytest.mark.parametrize("field", FIELDS)
def test_field_ops_random_vs_cpu(field):
    n = 1 << 16
    f = o.FIELDS[field]
    a = synth.field_elements(0xF1E1D + f.m % 97, n)
    b = synth.field_elements(0xF1E1E + f.m % 97, n)
    # make them proper residues (< m) -- the stream is < 2^254 < m already
    for op in ("add", "sub", "mul"):
        assert (api.field_op_dev(field, op, a, b) == cpu_ref.field_op(field, op, a, b)).all(), op
    for op in ("sqr", "neg", "from_mont", "to_mont"):
        assert (api.field_op_dev(field, op, a) == cpu_ref.field_op(field, op, a)).all(), op
    assert (api.field_op_dev(field, "inv", a[:2048]) == cpu_ref.field_op(field, "inv", a[:2048])).all()


@pytest.mark.parametrize("curve", CURVES)
def test_point_ops_golden(curve):
    kat = load_json("curve_kat.json")[curve]
    pj = unhex_rows([c["p_jac"] for c in kat["add_cases"]])
    qj = unhex_rows([c["q_jac"] for c in kat["add_cases"]])
    qa = unhex_rows([c["q_affine"] for c in kat["add_cases"]])
    want = unhex_rows([c["sum_affine"] for c in kat["add_cases"]])
    want_dbl = unhex_rows([c["dbl_p_affine"] for c in kat["add_cases"]])
    assert (api.point_op_dev(curve, "add", pj, qj)[:, :8] == want).all()
    assert (api.point_op_dev(curve, "madd", pj, qa)[:, :8] == want).all()
    assert (api.point_op_dev(curve, "dbl", pj)[:, :8] == want_dbl).all()


@pytest.mark.parametrize("curve", CURVES)
def test_bases_generate(curve):
    n = 3000
    b = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n, first=5)
    got = b.download()
    want = cpu_ref.gen_bases(curve, synth.BASE_S0 + 5 * synth.BASE_D, synth.BASE_D, n, threads=4)
    assert (got == want).all()


# ---------------------------------------------------------------------------------------
# K3: MSM
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("curve", CURVES)
def test_msm_golden_cases(curve):
    kat = load_json("msm_kat.json")[curve]
    for case in kat["cases"]:
        got = api.best_multiexp(curve, unhex_rows(case["scalars"]), unhex_rows(case["bases"]))
        assert (got[:8] == unhex(case["result_affine"])).all(), case["n"]
    for rec in kat["recipes"]:
        sc = synth.field_elements(rec["seed"], rec["n"])
        bases = api.Bases.generate(curve, rec["s0"], rec["d"], rec["n"])
        got = bases.msm(sc)
        assert (got[:8] == unhex(rec["result_affine"])).all(), rec["n"]


def test_msm_empty():
    got = api.best_multiexp("pallas", np.zeros((0, 4), np.uint64), np.zeros((0, 8), np.uint64))
    assert (got == 0).all()


def test_msm_length_mismatch_asserts():
    with pytest.raises(AssertionError):
        api.best_multiexp("pallas", np.zeros((3, 4), np.uint64), np.zeros((2, 8), np.uint64))


def _edge_inputs(curve, n, seed):
    """random scalars/bases with the reference-relevant edge cases mixed in"""
    f = o.CURVES[curve].scalar
    sc = synth.field_elements(seed, n)
    bases = cpu_ref.gen_bases(curve, 0xABCDEF + seed % 1000, 0x1357, n, threads=4)
    if n >= 16:
        sc[0] = 0                                                   # zero scalar (Montgomery 0)
        sc[1] = np.array(f.limbs(f.m - 1), np.uint64)               # -1
        sc[2] = np.array(f.limbs(1), np.uint64)                     # 1
        bases[3] = 0                                                # identity base
        bases[5] = bases[4]; sc[5] = sc[4]                          # duplicate pair, same scalar
        bases[7] = bases[6]; bases[7][4:] = cpu_ref.field_op(o.CURVES[curve].base.name, "neg", bases[6][4:].reshape(1, 4))[0]
        sc[7] = sc[6]                                               # P and -P cancel
        sc[9] = sc[2]; sc[10] = sc[2]; bases[10] = bases[9]         # 1*P + 1*P: doubling inside a bucket
    return sc, bases


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n", [1, 2, 5, 64, 1000, 4097, 1 << 14])
def test_msm_vs_cpu_ref(curve, n):
    sc, bases = _edge_inputs(curve, n, 0x5EED + n)
    want = aff(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=8))
    got = api.best_multiexp(curve, sc, bases)
    assert (got[:8] == want).all()
    one = np.array(o.CURVES[curve].base.limbs(1), np.uint64)
    assert (got[8:] == (one if want.any() else 0)).all()


@pytest.mark.parametrize("cbits", [2, 3, 5, 8, 11, 13, 15, 16, 17, 18])
def test_msm_window_widths(cbits):
    """every window width must give the same group element (exercises the signed recoding,
    including widths that divide 255 and leave a carry-only top window)"""
    curve, n = "pallas", 3000
    sc, bases = _edge_inputs(curve, n, 0xC0FFEE)
    want = aff(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=8))
    try:
        api.set_window_bits(cbits)
        got = api.best_multiexp(curve, sc, bases)
    finally:
        api.set_window_bits(0)
    assert (got[:8] == want).all()


def test_msm_canonical_scalars_and_offset():
    curve, n = "vesta", 2048
    sc, bases = _edge_inputs(curve, n, 0xCA11)
    can = cpu_ref.field_op("fp", "from_mont", sc)
    b = api.Bases.from_host(curve, bases)
    want = aff(curve, cpu_ref.best_multiexp(curve, sc[100:1100], bases[100:1100], threads=8))
    assert (b.msm(can[100:1100], offset=100, montgomery=False)[:8] == want).all()
    assert (b.msm(sc[100:1100], offset=100, montgomery=True)[:8] == want).all()
    with pytest.raises(api.TrhError):
        b.msm(sc, offset=1)  # range exceeds the resident bases


def test_msm_all_same_base_skewed():
    """all scalars small and all bases equal: every pair lands in a handful of buckets"""
    curve, n = "pallas", 5000
    f = o.CURVES[curve].scalar
    vals = [(i % 3) for i in range(n)]
    sc = np.array([f.limbs(v) for v in vals], np.uint64)
    g = np.array(o.CURVES[curve].affine_limbs(o.CURVES[curve].generator), np.uint64)
    bases = np.tile(g, (n, 1))
    got = api.best_multiexp(curve, sc, bases)
    want = o.CURVES[curve].mul(sum(vals), o.CURVES[curve].generator)
    assert o.CURVES[curve].affine_from_limbs(got[:8]) == want


@pytest.mark.parametrize("log_n", [16, 20, 24])
def test_msm_closed_form(log_n):
    """BASELINE config 2 (2^20 Pallas MSM) and the headline size 2^24: bases P_i = (s0 + i d) G so the expected result is
    (sum_i s_i (s0 + i d) mod q) G -- a size-independent check; at 2^16 also vs the CPU restatement."""
    curve = "pallas"
    n = 1 << log_n
    f = o.CURVES[curve].scalar
    sc = synth.msm_scalars(log_n)
    bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    got = bases.msm(sc)
    can = cpu_ref.field_op("fq", "from_mont", sc)
    total = synth.weighted_scalar_sum(can, synth.BASE_S0, synth.BASE_D) % f.m
    g = np.array(o.CURVES[curve].affine_limbs(o.CURVES[curve].generator), np.uint64)
    want = aff(curve, cpu_ref.scalar_mul(curve, g, np.array(o.int_to_limbs(total), np.uint64)))
    assert (got[:8] == want).all()
    if log_n <= 16:
        want2 = aff(curve, cpu_ref.best_multiexp(curve, sc, bases.download(), threads=8))
        assert (got[:8] == want2).all()


def test_msm_tiled_beyond_2_25():
    """above 2^25 pairs an MSM runs as range tiles whose points are added on the host (msm.hip MSM_TILE): closed form over a
    repeated 2^20 scalar block, ragged last tile, canonical scalars"""
    curve, BL, reps, extra = "vesta", 20, 32, 3
    q = o.CURVES[curve].scalar.m
    n = (reps << BL) + extra
    block = synth.field_elements(0x711ED, 1 << BL)  # < 2^254: canonical scalars
    T0 = synth.weighted_scalar_sum(block, 1, 0)
    T1 = synth.weighted_scalar_sum(block, 0, 1)
    total = reps * (synth.BASE_S0 * T0 + synth.BASE_D * T1) + synth.BASE_D * (1 << BL) * T0 * (reps * (reps - 1) // 2)
    total = (total + synth.weighted_scalar_sum(block[:extra], synth.BASE_S0, synth.BASE_D, start=reps << BL)) % q
    g = np.array(o.CURVES[curve].affine_limbs(o.CURVES[curve].generator), np.uint64)
    want = aff(curve, cpu_ref.scalar_mul(curve, g, np.array(o.int_to_limbs(total), np.uint64)))
    import torch
    bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    d = torch.from_numpy(block.view(np.int64)).cuda().repeat(reps + 1, 1)[:n].contiguous()
    got = bases.msm_dev(d, n, montgomery=False)
    assert (got[:8] == want).all()
    del d, bases
    torch.cuda.empty_cache()


def test_msm_batch_dev():
    curve, n, batch = "vesta", 1500, 5
    bases_h = cpu_ref.gen_bases(curve, 77, 3, n, threads=4)
    b = api.Bases.from_host(curve, bases_h)
    sc = synth.field_elements(0xBA7C4, n * batch).reshape(batch, n, 4)
    d = api.DeviceBuffer.from_host(sc)
    got = b.msm_batch_dev(d, n, batch)
    for k in range(batch):
        want = aff(curve, cpu_ref.best_multiexp(curve, sc[k], bases_h, threads=8))
        assert (got[k, :8] == want).all(), k


SMALL_SCRIPT = r"""
import numpy as np
import cpu_ref
from common import point_hex
from tiny_ram_halo2_amd import api, synth
api.init(0)
for curve, n, batch in (("pallas", 1, 1), ("vesta", 777, 1), ("pallas", 4098, 2), ("vesta", 8448, 1), ("pallas", 3000, 4)):
    b = api.Bases.from_host(curve, cpu_ref.gen_bases(curve, 91, 7, n, threads=4))
    sc = synth.field_elements(0x5A11 + n, n * batch).reshape(batch, n, 4)
    got = b.msm_batch_dev(api.DeviceBuffer.from_host(sc), n, batch)
    print("case", curve, n, batch, "".join(point_hex(got[k]) for k in range(batch)))
"""


def test_msm_small_kernel_boundaries_batches_and_plain_sums():
    """msm_small_kernel (one launch for MSMs of up to 8448 pairs in batches of up to four): the last size it takes and the first it does
    not, batches of 2 - 4 (an IPA round over the collapsed generators is a batch of two over 2^12 + 2 points), against the oracle; then
    the same cases in a child process with option reduce_q4 = 0 (the kernel's shuffle sums instead of its quad-lane sums): equal points"""
    from common import point_hex, run_with_options
    mine = {}
    for curve, n, batch in (("pallas", 1, 1), ("vesta", 777, 1), ("pallas", 4098, 2), ("vesta", 8448, 1), ("pallas", 3000, 4), ("vesta", 8449, 2)):
        bases_h = cpu_ref.gen_bases(curve, 91, 7, n, threads=4)
        b = api.Bases.from_host(curve, bases_h)
        sc = synth.field_elements(0x5A11 + n, n * batch).reshape(batch, n, 4)
        got = b.msm_batch_dev(api.DeviceBuffer.from_host(sc), n, batch)
        for k in range(batch):
            want = aff(curve, cpu_ref.best_multiexp(curve, sc[k], bases_h, threads=8))
            assert (got[k, :8] == want).all(), (curve, n, batch, k)
        mine[(curve, n, batch)] = "".join(point_hex(got[k]) for k in range(batch))
    out = run_with_options(SMALL_SCRIPT, {"TRH_REDUCE_Q4": "0"})
    seen = 0
    for line in out.splitlines():
        if line.startswith("case"):
            _, curve, n, batch, hexes = line.split()
            assert mine[(curve, int(n), int(batch))] == hexes, line[:40]
            seen += 1
    assert seen == 5


def test_point_sum_shards_equal_whole():
    """range-sharded MSM (the multi-GPU decomposition) == whole MSM"""
    curve, n, g = "pallas", 6000, 4
    sc, bases = _edge_inputs(curve, n, 0x5A4D)
    b = api.Bases.from_host(curve, bases)
    whole = b.msm(sc)
    per = n // g
    parts = np.stack([b.msm(sc[r * per:(r + 1) * per], offset=r * per) for r in range(g)])
    assert (api.point_sum(curve, parts) == whole).all()


# ---------------------------------------------------------------------------------------
# K4: NTT
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("field", FIELDS)
def test_ntt_golden(field):
    arr, meta = load_npz("ntt_kat.npz"), load_json("ntt_kat.json")
    for log_n in (0, 1, 2, 3, 4, 10):
        key = f"{field}_{log_n}"
        fwd = api.best_fft(field, arr[key + "_in"], unhex(meta[key]["omega"]), log_n)
        assert (fwd == arr[key + "_fwd"].reshape(-1, 4)).all(), key
        inv = api.best_fft(field, fwd, unhex(meta[key]["omega_inv"]), log_n)
        assert (inv == arr[key + "_inv_unscaled"].reshape(-1, 4)).all(), key


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("log_n", [1, 2, 5, 9, 11, 12, 13, 16, 18, 19, 20])
def test_ntt_vs_cpu_ref(field, log_n):
    f = o.FIELDS[field]
    a = synth.field_elements(0x4E5454 + log_n, 1 << log_n)
    w = np.array(f.limbs(f.omega(log_n)), np.uint64)
    got = api.best_fft(field, a, w, log_n)
    want = cpu_ref.best_fft(field, a, w, log_n, threads=8)
    assert (got == want).all()


def test_ntt_2_22_roundtrip_and_spot_values():
    """BASELINE config 3 (2^22 Fp NTT): bit-exact vs the CPU restatement, inverse round trip and
    Horner spot checks against the big-int oracle."""
    field, log_n = "fp", 22
    f = o.FIELDS[field]
    n = 1 << log_n
    a = synth.ntt_input(log_n)
    w = f.omega(log_n)
    wl = np.array(f.limbs(w), np.uint64)
    d = api.DeviceBuffer.from_host(a)
    api.ntt_dev(field, d, log_n, wl)
    fwd = d.to_host(shape=(-1, 4))
    assert (fwd == cpu_ref.best_fft(field, a, wl, log_n, threads=8)).all()
    coeffs = [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in a[:]]
    rinv = pow(f.R, -1, f.m)
    coeffs = [c * rinv % f.m for c in coeffs]
    for i in (0, 1, 3 * 2 ** 20 + 17):
        x, acc = pow(w, i, f.m), 0
        for cf in reversed(coeffs):
            acc = (acc * x + cf) % f.m
        assert f.from_limbs(fwd[i]) == acc
    api.ntt_dev(field, d, log_n, np.array(f.limbs(f.inv(w)), np.uint64))
    api.field_scale_dev(field, d, n, np.array(f.limbs(f.inv(n)), np.uint64))
    api.lib().trh_stream_synchronize(None)
    assert (d.to_host(shape=(-1, 4)) == a).all()


def test_ntt_batch_and_linearity():
    field, log_n, batch = "fq", 13, 6
    f = o.FIELDS[field]
    n = 1 << log_n
    a = synth.field_elements(0xBA7C5, n * batch).reshape(batch, n, 4)
    w = np.array(f.limbs(f.omega(log_n)), np.uint64)
    d = api.DeviceBuffer.from_host(a)
    api.ntt_dev(field, d, log_n, w, batch=batch)
    got = d.to_host(shape=(batch, n, 4))
    for k in range(batch):
        assert (got[k] == cpu_ref.best_fft(field, a[k], w, log_n, threads=8)).all(), k
    # linearity: NTT(a0 + a1) == NTT(a0) + NTT(a1)
    s = cpu_ref.field_op(field, "add", a[0], a[1])
    lhs = api.best_fft(field, s, w, log_n)
    rhs = cpu_ref.field_op(field, "add", got[0], got[1])
    assert (lhs == rhs).all()


def test_scale_periodic():
    field, n = "fp", 1000
    f = o.FIELDS[field]
    a = synth.field_elements(0x5CA1E, n)
    facs = np.array([f.limbs(1), f.limbs(f.ZETA), f.limbs(f.ZETA * f.ZETA % f.m)], np.uint64)
    d = api.DeviceBuffer.from_host(a)
    api.field_scale_periodic_dev(field, d, n, facs)
    api.lib().trh_stream_synchronize(None)
    want = cpu_ref.field_op(field, "mul", a, facs[np.arange(n) % 3])
    assert (d.to_host(shape=(-1, 4)) == want).all()


@pytest.mark.parametrize("curve", CURVES)
def test_msm_heavy_buckets(curve):
    """skewed scalars (values 0..3, as witness columns of flags and small words produce): a handful of
    buckets hold thousands of entries each and go through the workgroup-per-bucket combine"""
    n = 40000
    f = o.CURVES[curve].scalar
    vals = (np.arange(n) * 2654435761 >> 7) % 4
    table = np.array([f.limbs(v) for v in range(4)], np.uint64)
    sc = table[vals]
    bases = cpu_ref.gen_bases(curve, 0xFEED, 0x35, n, threads=4)
    want = aff(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=8))
    assert (api.best_multiexp(curve, sc, bases)[:8] == want).all()


def test_concurrent_callers():
    """entry points may be called from several host threads (rayon workers on the Rust side): results must not
    depend on the interleaving"""
    import threading
    f = o.FIELDS["fp"]
    log_n = 12
    w = np.array(f.limbs(f.omega(log_n)), np.uint64)
    inputs = [synth.field_elements(0x7EAD + t, 1 << log_n) for t in range(4)]
    sc, bases = _edge_inputs("vesta", 2000, 0x7EAD)
    want_fft = [cpu_ref.best_fft("fp", a, w, log_n, threads=2) for a in inputs]
    want_msm = aff("vesta", cpu_ref.best_multiexp("vesta", sc, bases, threads=4))
    errors = []

    def worker(t):
        try:
            for _ in range(5):
                assert (api.best_fft("fp", inputs[t], w, log_n) == want_fft[t]).all()
                assert (api.best_multiexp("vesta", sc, bases)[:8] == want_msm).all()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("kind,log_n", [("all_ones", 17), ("words16", 17), ("flags_quarter", 17), ("all_ones", 20), ("words16", 21), ("half_random", 20)])
def test_msm_witness_like_scalars(kind, log_n):
    """scalars shaped like TinyRAM witness columns (flags, small words, mostly-empty columns): millions of
    entries share a handful of buckets / one level-1 bin; exercises the chunk-parallel bucket sort (the fallback of the LDS
    bin sort when a bin does not fit) and the workgroup-per-bucket combine at sizes where they matter"""
    curve, n = "vesta", 1 << log_n
    cv = o.CURVES[curve]
    f = cv.scalar
    rng = np.random.default_rng(7)
    if kind == "all_ones":
        vals = np.ones(n, np.int64)
    elif kind == "words16":
        vals = rng.integers(0, 1 << 16, n)
    elif kind == "half_random":   # every second scalar is 3: one bin of the lowest window is oversize, all other windows see half-empty bins
        vals = np.where(np.arange(n) % 2 == 0, 3, rng.integers(0, 1 << 62, n))
    else:
        vals = np.where(np.arange(n) < n // 4, rng.integers(0, 2, n), 0)
    can = np.zeros((n, 4), np.uint64)      # canonical scalars: the values fit one limb
    can[:, 0] = vals.astype(np.uint64)
    sc = cpu_ref.field_op("fp" if curve == "vesta" else "fq", "to_mont", can)
    bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    got = bases.msm(sc)
    total = synth.weighted_scalar_sum(can, synth.BASE_S0, synth.BASE_D) % f.m
    g = np.array(cv.affine_limbs(cv.generator), np.uint64)
    want = aff(curve, cpu_ref.scalar_mul(curve, g, np.array(o.int_to_limbs(total), np.uint64)))
    assert (got[:8] == want).all()


BIN_SORT_SCRIPT = r"""
import numpy as np
from tiny_ram_halo2_amd import api, synth
from common import point_hex
api.init(0)
assert api.get_option("bin_sort") == 0
curve, n = "pallas", (1 << 20) + 7
bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
print("lone", point_hex(bases.msm_dev(api.DeviceBuffer.from_host(synth.field_elements(0xB175, n)), n)))
m = 1 << 16
small = api.Bases.generate(curve, 5, 9, m)
small.precompute(0)
print("batch", point_hex(small.msm_batch_dev(api.DeviceBuffer.from_host(synth.field_elements(0xB176, 4 * m)), m, 4)))
"""


def test_msm_bin_sort_equals_chunked_passes():
    """the whole-bin LDS bucket sort and the chunked passes it replaces (option bin_sort = 0, a fresh process: options are fixed while a
    context exists; also the fallback for oversize bins) sort the same entries: same point, single MSM and fixed-base batch"""
    from common import point_hex, run_with_options
    curve, n = "pallas", (1 << 20) + 7
    bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
    a = bases.msm_dev(api.DeviceBuffer.from_host(synth.field_elements(0xB175, n)), n)
    m = 1 << 16
    small = api.Bases.generate(curve, 5, 9, m)
    small.precompute(0)
    b = small.msm_batch_dev(api.DeviceBuffer.from_host(synth.field_elements(0xB176, 4 * m)), m, 4)
    out = dict(line.split() for line in run_with_options(BIN_SORT_SCRIPT, {"TRH_BIN_SORT": "0"}).splitlines() if line.startswith(("lone", "batch")))
    assert out["lone"] == point_hex(a) and out["batch"] == point_hex(b)


# ---------------------------------------------------------------------------------------
# fixed-base tables (trh_bases_precompute): same group elements as the per-window path and the oracle
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("cbits", [0, 6, 11, 16, 17, 18])
def test_msm_fixed_base_tables(curve, cbits):
    n = 3001
    sc, bases = _edge_inputs(curve, n, 0xF1BA5E + cbits)
    b = api.Bases.from_host(curve, bases)
    plain = b.msm(sc)
    used = b.precompute(cbits)
    assert used == (cbits or 9)  # automatic: log2(n) - 2
    want = aff(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=8))
    got = b.msm(sc)
    assert (got == plain).all() and (got[:8] == want).all()
    # a sub-range keeps using the per-window path over the same handle
    part = b.msm(sc[100:900], offset=100)
    assert (part[:8] == aff(curve, cpu_ref.best_multiexp(curve, sc[100:900], bases[100:900], threads=8))).all()


def test_msm_fixed_base_batch_and_skew():
    curve, n, batch = "vesta", (1 << 12) + 1, 6
    bases_h = cpu_ref.gen_bases(curve, 91, 5, n, threads=4)
    b = api.Bases.from_host(curve, bases_h)
    f = o.CURVES[curve].scalar
    sc = synth.field_elements(0xF1BA7C, n * batch).reshape(batch, n, 4).copy()
    sc[1, :] = np.array(f.limbs(1), np.uint64)          # all ones: one bucket of the lowest window holds everything
    sc[2, :] = np.array(f.limbs(f.m - 1), np.uint64)    # -1: carries ripple through every window
    sc[3, :] = 0
    sc[4, : n // 2] = np.array(f.limbs(3), np.uint64)
    d = api.DeviceBuffer.from_host(sc)
    plain = b.msm_batch_dev(d, n, batch)
    assert b.precompute(0) == 10
    got = b.msm_batch_dev(d, n, batch)
    assert (got == plain).all()
    for k in range(batch):
        assert (got[k, :8] == aff(curve, cpu_ref.best_multiexp(curve, sc[k], bases_h, threads=8))).all(), k


def test_msm_fixed_base_limits():
    b = api.Bases.generate("pallas", 5, 7, 1 << 10)
    with pytest.raises(api.TrhError):
        b.precompute(19)  # wider than the fixed-base sort supports
    w = api.Bases.wrap_device("pallas", api.lib().trh_bases_device_ptr(b.handle), 1 << 10)
    with pytest.raises(api.TrhError):
        w.precompute(0)  # wrapped memory is not immutable


def test_best_multiexp_host_cache():
    """the host-pointer entry point keeps recently seen base sets resident (pointer + length + fingerprint); results must not
    change over repeated calls (the fourth switches to fixed-base tables), and a different set at the same address is noticed"""
    curve, n = "vesta", 5000
    sc, bases = _edge_inputs(curve, n, 0xCAC4E)
    bases = np.ascontiguousarray(bases)
    want = aff(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=8))
    for rep in range(6):
        sc_r = np.ascontiguousarray(np.roll(sc, rep, axis=0))
        got = api.best_multiexp(curve, sc_r, bases)
        if rep == 0:
            assert (got[:8] == want).all()
        else:
            assert (got[:8] == aff(curve, cpu_ref.best_multiexp(curve, sc_r, bases, threads=8))).all(), rep
    # overwrite the SAME buffer with another set: the fingerprint must miss
    other = cpu_ref.gen_bases(curve, 0x5151, 7, n, threads=4)
    bases[:] = other
    got = api.best_multiexp(curve, sc, bases)
    assert (got[:8] == aff(curve, cpu_ref.best_multiexp(curve, sc, other, threads=8))).all()
