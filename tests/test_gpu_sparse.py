"""GPU tests (-m gpu) of the sparse-column path of batched fixed-base MSMs (csrc/msm.hip msm_sparse_*: Params::commit_lagrange over the
witness columns of the reference's circuit, reached from /root/reference/src/test_utils.rs:41-49).  Which path a column takes is decided
by its digits alone (sampler, then the emit's list counters); every case below is compared with the oracle's best_multiexp limb for
limb, so a column that is classified wrongly, emitted incompletely or handed to the wrong run of the dense pipeline shows up as a
different point.  Cases: the six value classes of the witness at k = 14 and k = 18, batches that mix sparse and full-size columns in
every order, the adversarial columns (one full-size scalar among flags; full-size values everywhere EXCEPT on the rows the sampler
looks at; all zeros; all ones; a column just below / above the list capacity), both curves."""
import numpy as np
import pytest
import torch

import cpu_ref
from tiny_ram_halo2_amd import api, replay, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init(0)
    yield


def mont(field, can):
    """canonical (.., 4) limbs -> Montgomery limbs (the prover's in-memory form)"""
    return cpu_ref.field_op(field, "to_mont", np.ascontiguousarray(can).reshape(-1, 4)).reshape(can.shape)


def commit_and_check(curve, k, cols_can, blinds, check_idx=None):
    """cols_can: (b, n, 4) canonical scalars; commits the batch over n + 1 generated bases with tables and compares the chosen
    columns (all by default) with cpu_ref.best_multiexp"""
    n = 1 << k
    sf = api.SCALAR_FIELD[curve]
    b = cols_can.shape[0]
    bases = api.Bases.generate(curve, synth.BASE_S0 + k, synth.BASE_D, n + 1)
    bases.precompute(0)
    cols = mont(sf, cols_can)
    d = torch.from_numpy(cols.view(np.int64)).cuda()
    got = bases.commit_batch_dev(d, n, b, blinds)
    xy = bases.download()
    th = cpu_ref.hardware_threads()
    for i in (range(b) if check_idx is None else check_idx):
        sc = np.concatenate([cols[i], blinds[i][None]])
        want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc, xy, threads=th))
        assert (np.asarray(got[i])[:8] == want).all(), (curve, k, i)
    bases.destroy()
    return got


def full_size(seed, count):
    return cpu_ref.field_op("fp", "from_mont", synth.field_elements(seed, count))  # uniformly random canonical values (either field: < 2^254)


@pytest.mark.parametrize("curve", ["vesta", "pallas"])
def test_witness_classes_k14(curve):
    """every value class of the witness (replay.witness_columns), blinded and not, in one batch of 24 at k = 14: all columns vs the oracle"""
    k, n = 14, 1 << 14
    parts = []
    for idx, (kind, blinded) in enumerate([("flag", False), ("word", False), ("flag", True), ("word", True), ("even", True), ("sorted", True), ("full", True), ("flag", True)]):
        parts.append(replay.witness_columns(kind, blinded, 0x5AA0 + idx, 3, n, 32))
    cols = np.concatenate(parts)
    commit_and_check(curve, k, cols, synth.field_elements(0xB1, cols.shape[0]))


def test_mixed_batches_and_adversarial_columns_k14():
    """sparse and dense columns interleaved (runs of one, runs at both ends), plus the columns meant to fool the classifier"""
    k, n = 14, 1 << 14
    flag = replay.witness_columns("flag", True, 1, 4, n, 32)
    word = replay.witness_columns("word", True, 2, 4, n, 32)
    full = replay.witness_columns("full", True, 3, 6, n, 32)
    zero = np.zeros((1, n, 4), dtype=np.uint64)
    ones = np.zeros((1, n, 4), dtype=np.uint64); ones[:, :, 0] = 1
    one_big = flag[:1].copy(); one_big[0, 77] = full_size(9, 1)[0]                    # one full-size scalar among flags: stays sparse
    # full-size values on every row EXCEPT the ones the sampler reads (multiples of n / 1024): looks empty, overflows the lists -> dense
    hidden = full_size(10, n).reshape(1, n, 4).copy(); hidden[0, :: n // 1024] = 0
    # half the rows full-size: about 11 digits each, far above the capacity W n / 8 -> dense by the sampler
    halfd = np.zeros((1, n, 4), dtype=np.uint64); halfd[0, ::2] = full_size(11, n // 2)
    # an eighth of the rows full-size: at the edge of the list capacity (either path must give the same point)
    edge = np.zeros((1, n, 4), dtype=np.uint64); edge[0, ::8] = full_size(12, n // 8)
    maxv = np.zeros((1, n, 4), dtype=np.uint64); maxv[0, : n // 4] = full_size(13, 1)[0]  # one repeated full-size value on the live rows
    for order in ([full[0:1], flag, full[1:3], word, zero, full[3:4], ones, one_big, hidden, halfd, edge, maxv, full[4:6]],
                  [flag, word, one_big, ones],                      # all sparse
                  [full, hidden, halfd],                            # all dense
                  [hidden, flag[:1], hidden, word[:1], hidden, zero, hidden, ones]):  # dense runs of length one between sparse columns
        cols = np.concatenate(order)
        commit_and_check("vesta", k, cols, synth.field_elements(0xB2 + cols.shape[0], cols.shape[0]))


def test_witness_classes_k18():
    """the real size: a batch of 64 at k = 18 with the proof's class mix; one column of every class, the adversarial ones and two
    full-size columns against the oracle (an MSM of 2^18 + 1 pairs on the CPU takes a fraction of a second)"""
    k, n = 18, 1 << 18
    blocks = [replay.witness_columns("flag", False, 21, 10, n, 32), replay.witness_columns("word", False, 22, 6, n, 32),
              replay.witness_columns("flag", True, 23, 16, n, 32), replay.witness_columns("full", True, 24, 3, n, 32),
              replay.witness_columns("word", True, 25, 10, n, 32), replay.witness_columns("even", True, 26, 6, n, 32),
              replay.witness_columns("sorted", True, 27, 8, n, 32), replay.witness_columns("full", True, 28, 3, n, 32)]
    cols = np.concatenate(blocks)
    one_big = cols[16:17].copy(); one_big[0, 12345] = full_size(31, 1)[0]
    hidden = full_size(32, n).reshape(1, n, 4).copy(); hidden[0, :: n // 1024] = 0
    cols = np.concatenate([cols, one_big, hidden])
    assert cols.shape[0] == 64
    got = commit_and_check("vesta", k, cols, synth.field_elements(0xB3, 64), check_idx=[0, 10, 16, 31, 32, 35, 45, 51, 59, 60, 62, 63])
    # the same columns one at a time through the single-MSM entry (its own path decisions): the same points
    bases = api.Bases.generate("vesta", synth.BASE_S0 + k, synth.BASE_D, n + 1)
    bases.precompute(0)
    colsm = mont("fp", cols[[0, 16, 45, 63]])
    bl = synth.field_elements(0xB3, 64)[[0, 16, 45, 63]]
    for j, i in enumerate([0, 16, 45, 63]):
        sc = torch.from_numpy(np.concatenate([colsm[j], bl[j][None]]).view(np.int64)).cuda()
        assert (bases.msm_dev(sc, n + 1) == got[i]).all(), i
    bases.destroy()


def test_unit_digits_and_tiny_columns_k14():
    """a chunk whose sampled rows show nothing but 0 / +-1 digits takes the unit path: digits equal to +-1 are summed from the table directly
    and a column with at most 256 other digits skips the sort (msm_unit_sum_kernel, msm_unit_final_kernel); columns with more digits than the
    sampler saw go through the compact pipeline and get their unit sums added.  +-1 digits in every window (2^16 j, 2^16 j - 1 -> -1 then a carry), all-ones, flags with 0 / a few / exactly
    as many / one more than the tiny limit of other digits, chunks without any general column (no pipeline launch at all) and chunks that
    mix all three classes, repeated and identity bases under the ones"""
    k, n = 14, 1 << 14
    def flags(seed, density=0.5):
        f = np.zeros((1, n, 4), dtype=np.uint64)
        f[0, : n // 4, 0] = (np.random.default_rng(seed).random(n // 4) < density).astype(np.uint64)
        return f
    ones = np.zeros((1, n, 4), dtype=np.uint64); ones[:, :, 0] = 1
    zero = np.zeros((1, n, 4), dtype=np.uint64)
    # powers of two at window boundaries and just below them, for every plausible window width: +1 digits in high windows, -1 with a carry
    pw = np.zeros((1, n, 4), dtype=np.uint64)
    vals = []
    for w in range(10, 18):
        for j in range(1, 6):
            vals += [1 << (w * j), (1 << (w * j)) - 1, (1 << (w * j)) + 1, (1 << (w * j)) - 2]
    for t, v in enumerate(vals):
        for limb in range(4):
            pw[0, 3 * t + 1, limb] = (v >> (64 * limb)) & 0xFFFFFFFFFFFFFFFF
    def with_big(base, count, seed):
        """`count` full-size scalars (about W other digits each) on rows the sampler does NOT read (it reads every 16th row at k = 14): the
        chunk still votes for the unit path and the column's class is decided by its real digit count (tiny up to 256 entries, general above)"""
        col = base.copy()
        if count:
            rows = 16 * np.random.default_rng(seed).choice(n // 16, size=count, replace=False) + 5
            col[0, rows] = full_size(seed, count)
        return col
    f0 = flags(1)
    cases = [f0, with_big(f0, 3, 2), with_big(f0, 14, 3), with_big(f0, 15, 4), with_big(f0, 16, 5), with_big(f0, 17, 6), with_big(f0, 40, 7),
             ones, zero, pw, with_big(ones, 5, 8), flags(9, 0.02), flags(10, 1.0)]
    word = replay.witness_columns("word", True, 11, 2, n, 32)
    full = replay.witness_columns("full", True, 12, 2, n, 32)
    for order in (cases[:8],                                           # tiny (and nearly tiny) columns only
                  [ones, zero, pw, f0, f0, f0, f0, f0],                # nothing but unit entries and a handful of digits: no pipeline launch
                  cases + [word[0:1], full[0:1], word[1:2], full[1:2]]):  # all three classes in one chunk
        cols = np.concatenate(order)
        commit_and_check("vesta", k, cols, synth.field_elements(0xB7 + cols.shape[0], cols.shape[0]))
    # a lone commitment is a chunk of one (trh_msm_dev over the tabled set): every case on its own against the oracle
    bases = api.Bases.generate("vesta", synth.BASE_S0 + k, synth.BASE_D, n + 1)
    bases.precompute(0)
    xy = bases.download()
    blind = synth.field_elements(0xB9, 1)
    for i, col in enumerate(cases):
        sc = np.concatenate([mont("fp", col)[0], blind])
        got1 = bases.msm_dev(torch.from_numpy(sc.view(np.int64)).cuda(), n + 1)
        want = cpu_ref.to_affine("vesta", cpu_ref.best_multiexp("vesta", sc, xy, threads=cpu_ref.hardware_threads()))
        assert (np.asarray(got1)[:8] == want).all(), i
    bases.destroy()
    # zero blinds as well: a column of zeros then commits to the identity, whichever path it takes
    cols = np.concatenate([zero, f0, zero, ones, zero, zero, f0, zero])
    got = commit_and_check("pallas", k, cols, np.zeros((8, 4), dtype=np.uint64), check_idx=[1, 3, 6])
    bases = api.Bases.generate("pallas", synth.BASE_S0 + k, synth.BASE_D, n + 1)
    bases.precompute(0)
    lone = bases.msm_dev(torch.zeros((n + 1, 4), dtype=torch.int64, device="cuda"), n + 1)
    for i in (0, 2, 4, 5, 7):
        assert (got[i] == lone).all(), i
    bases.destroy()
