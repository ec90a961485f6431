"""GPU test (-m gpu): a bounded randomised soak under the driver's clock -- tools/soak.py's trial mix over the
MSM (random sizes 1 .. 2^19, window widths, split points, fixed-base tables, batches, identity / duplicate bases, small / equal / edge
scalars), the NTT / EvaluationDomain round trips, the lookup permutation, the coset-block domain, the host-pointer entries and the
product columns, and round 4's paths: batches that mix sparse and full-size columns against lone commitments, zero-padded best_fft with
stray non-zero elements in the padding, range-sharded MSMs over the device group with host / page-locked / device scalars and the forced
no-peer hand-over; round 6: IPA openings over tabled sets (small sets through msm_small_kernel, larger ones with their generators
collapsed -- csrc/ipafold.hip) against the same openings without tables.  Different code paths of libtrh must agree bit for bit, and the MSM (n <= 2^16) and NTT (k <= 18) trials are also
compared with the oracle's best_multiexp / best_fft (/root/reference reaches them through src/test_utils.rs:41-49).
The seed is fixed and printed; TRH_SOAK_SEED / TRH_SOAK_SECONDS / TRH_SOAK_KINDS override it for a longer or narrower run."""
import os
import sys

import pytest

import cpu_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bounded_soak():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import soak
    seed = int(os.environ.get("TRH_SOAK_SEED", "20261003"))
    seconds = float(os.environ.get("TRH_SOAK_SECONDS", "90"))
    kinds = [k for k in os.environ.get("TRH_SOAK_KINDS", "").split(",") if k] or None  # e.g. TRH_SOAK_KINDS=sparse: one kind of trial only
    stats, fails = soak.run(seconds, seed, oracle=cpu_ref, kinds=kinds)
    print("soak:", stats, "failures:", fails)
    assert not fails, f"seed {seed}: mismatches in {fails} ({stats})"
    # every kind of trial ran, and the oracle saw a share of them
    assert all(stats[k] > 0 for k in (kinds or ("msm", "ntt", "lookup", "blocks", "hostio", "products", "sparse", "padded", "sharded", "opening"))), (seed, stats)
    assert stats["vs_oracle"] > 0 or (kinds and not {"msm", "ntt"} & set(kinds))
