"""GPU parity tests (-m gpu) at the sizes the prover really runs (BASELINE configs 2, 4, 5): the paths that round 1 only
timed or self-checked are compared with the C++ oracle (oracle/cpu_ref.cpp, pinned to the big-int restatement at small k in
tests/test_oracle.py):

  * trh_ipa_create_proof transcript identity + the verifier equation at k = 10, 14, 16, 18 (the window-override path k - 8 of
    csrc/ipa.hip, the G = 4 combine, the pinned-ring tail writes);
  * poly::multiopen::create_proof at k = 10 and 12;
  * trh_point_fft_dev at k = 10, 14 (literal restatement) and 18 (closed form over known discrete logs + definition-level MSM
    spot checks over unstructured bases);
  * 2^20 Pallas / 2^18 + 1 Vesta MSMs limb-for-limb against best_multiexp with UNSTRUCTURED bases (hashed discrete logs);
  * a 2^26 Pallas MSM as 8 logical range shards on one device through the C ABI's device group (config 5's decomposition).

Reference call sites: /root/reference/src/test_utils.rs:20-25, 41-49."""
import random

import numpy as np
import pytest
import torch

import cpu_ref
import pasta as o
from common import LimbTranscript, run_with_options, ipa_verify_fast, multiopen_create_proof_fast
from tiny_ram_halo2_amd import api, ipa, multiopen, poly, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init(0)
    yield


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64).copy()).cuda()


def to_host(t):
    torch.cuda.synchronize()
    return t.contiguous().cpu().numpy().view(np.uint64)


class IntTranscript(LimbTranscript):
    """the device-side mirrors (ipa.py / multiopen.py) take challenges as canonical ints; same bytes, same hash"""

    def __init__(self, field):
        super().__init__(field)
        self.points, self.challenges = [], []

    def write_point(self, jac):
        self.points.append(np.ascontiguousarray(jac, dtype=np.uint64)[:8].copy())
        super().write_point(jac)

    def squeeze_challenge_scalar(self):
        c = self.field.from_limbs(super().squeeze_challenge_scalar())
        self.challenges.append(c)
        return c


def _params(curve, k, seed, precompute=False):
    n = 1 << k
    g_l = cpu_ref.gen_bases_hashed(curve, seed, n)          # unstructured generators
    w_l = cpu_ref.gen_bases_hashed(curve, seed ^ 0x5151, 1)
    u_l = cpu_ref.gen_bases_hashed(curve, seed ^ 0x6262, 1)
    return g_l, w_l, u_l, poly.Params(curve, k, g_l, g_l, w_l, u=u_l, precompute=precompute)


@pytest.mark.parametrize("curve,k", [("vesta", 10), ("pallas", 10), ("vesta", 14), ("pallas", 14), ("vesta", 16), ("pallas", 16), ("vesta", 18), ("pallas", 18)])
def test_ipa_native_vs_cpp_oracle(curve, k):
    """the single-call prover against the literal C++ restatement (G' collapsed with scalar multiplications there, never
    materialised here): identical S, L_j, R_j, c, f; then the verifier equation on what the GPU produced"""
    _ipa_case(curve, k, precompute=False)


@pytest.mark.parametrize("curve,k", [("vesta", 10), ("pallas", 10), ("vesta", 14), ("pallas", 15), ("pallas", 16), ("vesta", 16), ("pallas", 17), ("vesta", 18)])
def test_ipa_native_fixed_base_tables_vs_cpp_oracle(curve, k):
    """the same with Params that carry fixed-base tables: the opening then runs over ONE resident set g || w || u with its own table
    (poly.Params.ipa_bases -> the n + 2 form of trh_ipa_create_proof) and every MSM of it in fixed-base mode.  Up to k = 13 the set is small enough for msm_small_kernel and the opening leaves the
    table alone (level 0 of it serves as the base records); from k = 14 the generators are collapsed to 2^12 after k - 12 rounds
    (csrc/ipafold.hip: table windows of 12 .. 16 bits = sub-digits of 6 + 6 .. 8 + 8 bits) and the other rounds run over the 2^12 + 2 collapsed
    points through msm_small_kernel: every L_j, R_j must still be the literal prover's"""
    _ipa_case(curve, k, precompute=True)


def _ipa_case(curve, k, precompute):
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    g_l, w_l, u_l, params = _params(curve, k, 0x1FA0 + k, precompute)
    if precompute:
        assert len(params.ipa_bases()) == n + 2 and int(api.lib().trh_bases_precomputed_window_bits(params.ipa_bases().handle)) > 0
    p_l, s_l = synth.field_elements(0xA000 + k, n), synth.field_elements(0xB000 + k, n)
    rnd = random.Random(0x1FA + k)
    p_blind, s_blind, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
    draws = [rnd.randrange(fs.m) for _ in range(2 * k)]
    lim = lambda v: np.array(fs.limbs(v), np.uint64)  # noqa: E731

    it_dev = iter(draws)
    t_dev = IntTranscript(fs)
    collapses, small = api.stat("ipa_generator_collapses"), api.stat("msm_small_launches")
    c_dev, f_dev = ipa.create_proof_native(params, lambda: next(it_dev), t_dev, to_dev(p_l), p_blind, x3, s_l, s_blind)
    if precompute and api.get_option("ipa_fold") == 1:   # the paths the docstrings name were taken, not silently fallen back from
        assert api.stat("ipa_generator_collapses") - collapses == (1 if k >= 14 else 0)
        assert api.stat("msm_small_launches") - small == (12 if k >= 14 else k + 1)   # the rounds over 2^12 + 2 points / every MSM of a small opening
    it_ref = iter(draws)
    t_ref = LimbTranscript(fs)
    c_ref, f_ref = cpu_ref.ipa_create_proof(curve, k, g_l, w_l[0], u_l[0], lambda: lim(next(it_ref)), t_ref, p_l, lim(p_blind), lim(x3), s_l, lim(s_blind))
    assert (c_dev, f_dev) == (fs.from_limbs(c_ref), fs.from_limbs(f_ref))
    assert len(t_dev.log) == 1 + 2 * k + 2
    for i, (a, b) in enumerate(zip(t_dev.log, t_ref.log)):
        assert a == b, f"transcript item {i}"

    # verifier: P = commit(p, p_blind) on the device; v = p(x3) by the oracle
    commitment = params.commit(p_l, lim(p_blind))[:8]
    v = fs.from_limbs(cpu_ref.eval_polynomial(api.SCALAR_FIELD[curve], p_l, lim(x3)))
    xi, z, ch = t_dev.challenges[0], t_dev.challenges[1], t_dev.challenges[2:]
    rounds = [(t_dev.points[1 + 2 * j], t_dev.points[2 + 2 * j]) for j in range(k)]
    args = (curve, k, g_l, w_l[0], u_l[0], commitment, x3)
    tail = (t_dev.points[0], xi, z, rounds, ch)
    assert ipa_verify_fast(*args, v, *tail, c_dev, f_dev)
    if k <= 14:
        assert not ipa_verify_fast(*args, (v + 1) % fs.m, *tail, c_dev, f_dev)


def test_ipa_skewed_polynomial_takes_the_chunked_sort_again():
    """The opening's round MSMs promise the sort uniformly random scalars (dense_hint) and skip the launches of the chunked fallback behind
    the whole-bin LDS sort; msm_finish checks the sort's "a bin did not fit" flags and repeats the MSM with the fallback when one is set.
    A polynomial whose coefficients are all the SAME value with the digit 1 in every window of the table (and s(X) = 0) puts every entry
    of round 0 into bucket 1 of one bin -- far above the LDS capacity: the retry must happen (trh_stat msm_lean_retries) and the transcript
    must still be the oracle's, point for point."""
    if api.get_option("bin_sort") == 0:
        pytest.skip("option bin_sort = 0: every MSM takes the chunked passes, there is no lean sort to fall back from")
    curve, k = "pallas", 14
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    g_l, w_l, u_l, params = _params(curve, k, 0x5E3D, precompute=True)
    cbits = int(api.lib().trh_bases_precomputed_window_bits(params.ipa_bases().handle))
    assert len(params.ipa_bases()) == n + 2 and cbits > 0
    c0 = sum(1 << (cbits * j) for j in range(255 // cbits + 1)) % fs.m   # digit 1 in every window, no carries
    lim = lambda v: np.array(fs.limbs(v), np.uint64)  # noqa: E731
    p_l = np.tile(lim(c0), (n, 1))
    s_l = np.zeros((n, 4), dtype=np.uint64)
    rnd = random.Random(0x5E3D)
    p_blind, s_blind, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
    draws = [rnd.randrange(fs.m) for _ in range(2 * k)]
    before = api.stat("msm_lean_retries")
    it_dev = iter(draws)
    t_dev = IntTranscript(fs)
    c_dev, f_dev = ipa.create_proof_native(params, lambda: next(it_dev), t_dev, to_dev(p_l), p_blind, x3, s_l, s_blind)
    assert api.stat("msm_lean_retries") > before, "the skewed round did not overflow a bin: the test no longer reaches the retry"
    it_ref = iter(draws)
    t_ref = LimbTranscript(fs)
    c_ref, f_ref = cpu_ref.ipa_create_proof(curve, k, g_l, w_l[0], u_l[0], lambda: lim(next(it_ref)), t_ref, p_l, lim(p_blind), lim(x3), s_l, lim(s_blind))
    assert (c_dev, f_dev) == (fs.from_limbs(c_ref), fs.from_limbs(f_ref))
    for i, (a, b) in enumerate(zip(t_dev.log, t_ref.log)):
        assert a == b, f"transcript item {i}"


@pytest.mark.parametrize("curve,k", [("vesta", 10), ("pallas", 12), ("vesta", 18)])
def test_multiopen_vs_cpp_oracle(curve, k):
    """poly::multiopen::create_proof on resident polynomials, transcript-identical to the oracle's restatement; k = 18 over Vesta is
    the reference's own configuration (/root/reference/src/test_utils.rs:20-21, 41-49)"""
    cv = o.CURVES[curve]
    fs = cv.scalar
    n = 1 << k
    g_l, w_l, u_l, params = _params(curve, k, 0x0BE0 + k)
    rnd = random.Random(0x0BE1 + k)
    keys = ["a", "b", "c", "z", "h"]
    polys = {key: [rnd.randrange(fs.m) for _ in range(n)] for key in keys}
    blinds = {key: rnd.randrange(fs.m) for key in keys}
    x = rnd.randrange(fs.m)
    wgen = fs.omega(k)
    xw, xwinv = x * wgen % fs.m, x * pow(wgen, -1, fs.m) % fs.m
    queries = [(x, "a"), (x, "b"), (xw, "b"), (xwinv, "b"), (x, "c"), (x, "z"), (xw, "z"), (x, "h"), (xw, "b")]
    draws = [rnd.randrange(fs.m) for _ in range(2 + n + 1 + 2 * k)]
    it1, it2 = iter(draws), iter(draws)
    t_dev, t_ref = IntTranscript(fs), LimbTranscript(fs)
    dev_polys = {key: to_dev(np.array([fs.limbs(v) for v in polys[key]], np.uint64)) for key in keys}
    got = multiopen.create_proof(params, lambda: next(it1), t_dev, queries, dev_polys, blinds)
    want = multiopen_create_proof_fast(curve, k, g_l, w_l[0], u_l[0], lambda: next(it2), t_ref, queries, polys, blinds)
    assert got == want
    assert len(t_dev.log) == 1 + 3 + (1 + 2 * k + 2)
    assert t_dev.log == t_ref.log


@pytest.mark.parametrize("curve,k", [("vesta", 10), ("pallas", 10), ("vesta", 14)])
def test_point_fft_vs_cpp_oracle(curve, k):
    """Params::new's g -> g_lagrange step (best_fft over curve points, x n^-1, batch_normalize) over unstructured points with an
    identity among them, every output compared with the literal restatement"""
    fs = o.CURVES[curve].scalar
    n = 1 << k
    g = cpu_ref.gen_bases_hashed(curve, 0xF00 + k, n)
    g[5] = 0
    got = to_host(poly.Params.g_lagrange_from_g(curve, k, to_dev(g)))
    lim = lambda v: np.array(fs.limbs(v), np.uint64)  # noqa: E731
    want = cpu_ref.best_fft_points(curve, g, lim(fs.inv(fs.omega(k))), k)
    want = cpu_ref.scale_points_each(curve, want, np.tile(lim(fs.inv(1 << k)), (n, 1)))
    assert (got == want).all()


def test_point_fft_k18():
    """k = 18 (the reference's Params::new(18), /root/reference/src/test_utils.rs:20-21): the literal restatement needs 2.4 M
    scalar multiplications, so (i) inputs with known discrete logs a_j = s_j G: every output must be n^-1 NTT(s)_i G with the
    NTT over the scalar field by the oracle's best_fft, and (ii) unstructured inputs: sampled outputs against the definition
    out_i = n^-1 sum_j omega^-ij a_j evaluated by the oracle's best_multiexp"""
    curve, k = "vesta", 18
    cv = o.CURVES[curve]
    fs, sf = cv.scalar, api.SCALAR_FIELD[curve]
    n = 1 << k
    lim = lambda v: np.array(fs.limbs(v), np.uint64)  # noqa: E731
    w_inv, n_inv = fs.inv(fs.omega(k)), fs.inv(n)
    # (i) structured
    logs = cpu_ref.hashed_scalars(0x18F, n)                      # canonical h_j
    g = cpu_ref.gen_bases_hashed(curve, 0x18F, n)                # h_j G
    got = to_host(poly.Params.g_lagrange_from_g(curve, k, to_dev(g)))
    ntt = cpu_ref.best_fft(sf, cpu_ref.field_op(sf, "to_mont", logs), lim(w_inv), k, threads=cpu_ref.hardware_threads())
    ntt = cpu_ref.field_op(sf, "mul", ntt, np.tile(lim(n_inv), (n, 1)))
    gen = np.array(cv.affine_limbs(cv.generator), np.uint64)
    want = cpu_ref.scale_points(curve, gen, ntt)
    assert (got == want).all()
    # (ii) definition-level spot checks on the same transform: row i is an MSM with scalars n^-1 omega^(-i j)
    for i in (0, 1, n // 2 + 3, n - 1):
        step = pow(w_inv, i, fs.m)
        pw = np.tile(lim(n_inv), (n, 1))
        span, cur = 1, step
        while span < n:
            pw[span:2 * span] = cpu_ref.field_op(sf, "mul", pw[:span], np.tile(lim(cur), (span, 1)))
            cur, span = cur * cur % fs.m, span * 2
        row = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, pw, g, threads=cpu_ref.hardware_threads()))
        assert (got[i] == row).all(), i


@pytest.mark.parametrize("curve,n", [("pallas", 1 << 20), ("vesta", (1 << 18) + 1)])
def test_msm_unstructured_bases_vs_best_multiexp(curve, n):
    """BASELINE config 2 ("2^20 Pallas MSM ... bit-exact vs best_multiexp") and the prover's own size 2^18 + 1: random
    254-bit scalars, bases with hashed discrete logs (no arithmetic progression), limb-for-limb against the oracle's
    best_multiexp -- through the resident-bases entry, the host-pointer drop-in and the device-scalar entry"""
    sc = synth.field_elements(0x5CA1 + n, n)
    bases = cpu_ref.gen_bases_hashed(curve, 0xBA5E + n, n)
    want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=cpu_ref.hardware_threads()))
    b = api.Bases.from_host(curve, bases)
    assert (b.msm(sc)[:8] == want).all()
    assert (b.msm_dev(to_dev(sc), n)[:8] == want).all()
    assert (api.best_multiexp(curve, sc, bases)[:8] == want).all()
    b.destroy()


def test_msm_2_24_unstructured_bases_vs_best_multiexp():
    """north_star's target size: 2^24 Pallas pairs, random 254-bit scalars, UNSTRUCTURED bases (hashed discrete logs: no arithmetic
    progression a closed form could hide behind), limb-for-limb against the oracle's best_multiexp -- through
    the resident-bases entry with device scalars (the headline path) and through the tiled host-pointer drop-in"""
    curve, n = "pallas", 1 << 24
    threads = cpu_ref.hardware_threads()
    sc = synth.msm_scalars(24)
    bases = cpu_ref.gen_bases_hashed(curve, 0xBA5E24, n, threads)
    want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=threads))
    assert want.any()
    b = api.Bases.from_host(curve, bases)
    assert (b.msm_dev(to_dev(sc), n)[:8] == want).all()
    b.destroy()
    assert (api.best_multiexp(curve, sc, bases)[:8] == want).all()


def test_msm_2_26_as_8_logical_shards():
    """BASELINE config 5's decomposition on the one GPU of the test box: a device group of 8 contexts on device 0, the 2^26 bases
    range-sharded 8 x 2^23 by trh_bases_generate, device-resident scalars handed to the shards, 8 local Pippengers, the 8 partial
    points added on the host -- all inside the C ABI; closed form (the CPU restatement would take minutes)"""
    curve, log_n, G = "pallas", 26, 8
    n = 1 << log_n
    f = o.CURVES[curve].scalar
    api.init_multi([0] * G)
    api.set_shard_min(1 << 20)
    try:
        assert api.group_size() == G
        bases = api.Bases.generate(curve, synth.BASE_S0, synth.BASE_D, n)
        assert bases.shards() == G
        block = synth.field_elements(0x26C0DE, 1 << 22)  # canonical scalars < 2^254, repeated 16 times
        reps = n >> 22
        T0 = synth.weighted_scalar_sum(block, 1, 0)
        T1 = synth.weighted_scalar_sum(block, 0, 1)
        total = (reps * (synth.BASE_S0 * T0 + synth.BASE_D * T1) + synth.BASE_D * (1 << 22) * T0 * (reps * (reps - 1) // 2)) % f.m
        g = np.array(o.CURVES[curve].affine_limbs(o.CURVES[curve].generator), np.uint64)
        want = cpu_ref.to_affine(curve, cpu_ref.scalar_mul(curve, g, np.array(o.int_to_limbs(total), np.uint64)))
        d = torch.from_numpy(block.view(np.int64)).cuda().repeat(reps, 1).contiguous()
        got = bases.msm_dev(d, n, montgomery=False)
        assert (got[:8] == want).all()
        del d
        bases.destroy()
    finally:
        api.set_shard_min(1 << 62)  # later tests of the session create single-device sets again
        torch.cuda.synchronize()


FOLD_SCRIPT = r"""
import hashlib, random
import numpy as np, torch
import cpu_ref
from tiny_ram_halo2_amd import api, ipa, poly, synth
from common import DeviceTranscript
import pasta as o
api.init(0)
curve, k = "pallas", 16
fs = o.CURVES[curve].scalar
n = 1 << k
g_l = cpu_ref.gen_bases(curve, 29, 13, n, threads=8)
w_l = cpu_ref.gen_bases(curve, 515152, 1, 1, threads=1)
u_l = cpu_ref.gen_bases(curve, 626263, 1, 1, threads=1)
params = poly.Params(curve, k, g_l, g_l, w_l, u=u_l, precompute=True)
params.reserve(2)
p_l, s_l = synth.field_elements(0xF01D, n), synth.field_elements(0xF01E, n)
rnd = random.Random(0xF01D)
p_blind, s_blind, x3 = rnd.randrange(fs.m), rnd.randrange(fs.m), rnd.randrange(fs.m)
draws = [rnd.randrange(fs.m) for _ in range(2 * k)]
p_dev = torch.from_numpy(np.ascontiguousarray(p_l, dtype=np.uint64).view(np.int64).copy()).cuda()
for rep in range(2):   # twice: the second opening reuses the collapse's buffers and streams
    it = iter(draws)
    tr = DeviceTranscript(fs.m)
    c, f = ipa.create_proof_native(params, lambda: next(it), tr, p_dev, p_blind, x3, s_l, s_blind)
    h = hashlib.sha256(repr((c, f, tr.log)).encode()).hexdigest()
    print("digest", h)
print("fold", api.get_option("ipa_fold"))
"""


def test_ipa_generator_collapse_levels_agree():
    """option ipa_fold (csrc/ipa.hip: the round after which the generators are collapsed; 1 = the library's choice): never, the default, after
    4 / 7 / 8 rounds (collapsed sets of 2^12, 2^9, 2^8 points here) -- the same transcript, twice in every process (the second opening reuses
    the collapse's buffers).  (The default is checked against the literal prover above.)"""
    digests = {}
    for fold in ("0", "1", "4", "7", "8"):
        out = run_with_options(FOLD_SCRIPT, {"TRH_IPA_FOLD": fold}, timeout=900)
        d = [line.split()[1] for line in out.splitlines() if line.startswith("digest")]
        assert len(d) == 2 and d[0] == d[1], (fold, d)
        assert ("fold " + fold) in out
        digests[fold] = d[0]
    assert len(set(digests.values())) == 1, digests
