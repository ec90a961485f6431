"""The quad-lane group law (csrc/curve_q4.h: one XYZZ point over the four lanes of a DPP quad) against the golden vectors and the C++ oracle, and the
bucket reduction built on it (msm_reduce_q4_kernel) against the one-thread-per-slice kernels and the oracle.  /root/reference/src/test_utils.rs:41-49
reaches this arithmetic through best_multiexp's bucket sums (halo2_proofs 0.2.0 arithmetic.rs) and pasta_curves' Point addition."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cpu_ref
from common import load_json, unhex_rows
from tiny_ram_halo2_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CURVES = ["pallas", "vesta"]


@pytest.fixture(scope="module", autouse=True)
def _init():
    api.init(0)
    yield


@pytest.mark.parametrize("curve", CURVES)
def test_q4_point_ops_golden_and_edge_cases(curve):
    kat = load_json("curve_kat.json")[curve]
    pj = unhex_rows([c["p_jac"] for c in kat["add_cases"]])
    qj = unhex_rows([c["q_jac"] for c in kat["add_cases"]])
    want = unhex_rows([c["sum_affine"] for c in kat["add_cases"]])
    want_dbl = unhex_rows([c["dbl_p_affine"] for c in kat["add_cases"]])
    assert (api.point_op_dev(curve, "q4_add", pj, qj)[:, :8] == want).all()
    assert (api.point_op_dev(curve, "q4_dbl", pj)[:, :8] == want_dbl).all()
    # P + P, P + (-P), identity on either side, identity + identity -- in one launch, so that the rare branches run beside ordinary quads
    gen = api.Bases.generate(curve, 3, 5, 64).download()            # affine (x, y)
    one = api.point_op_dev(curve, "madd", np.zeros((1, 12), np.uint64), gen[:1])[0, 8:]  # 0 + G_0 = (x, y, 1): the Montgomery one
    p = np.zeros((64, 12), dtype=np.uint64)
    p[:, :8] = gen
    p[:, 8:] = one
    q = np.roll(p, 1, axis=0).copy()                   # ordinary pairs at rows 0 mod 4
    q[1::4] = p[1::4]                                  # P + P
    q[2::4] = 0                                        # P + identity
    pz = p.copy()
    pz[3::4] = 0                                       # identity + Q
    assert (api.point_op_dev(curve, "q4_add", pz, q) == api.point_op_dev(curve, "add", pz, q)).all()
    both = np.zeros((8, 12), dtype=np.uint64)
    assert (api.point_op_dev(curve, "q4_add", both, both) == 0).all() and (api.point_op_dev(curve, "q4_dbl", both) == 0).all()
    # P + (-P) in every quad of the launch, and P + P in every quad
    negy = p.copy()
    negy[:, 4:8] = cpu_ref.field_op({"pallas": "fp", "vesta": "fq"}[curve], "neg", p[:, 4:8].copy())
    assert (api.point_op_dev(curve, "q4_add", p, negy) == 0).all()
    assert (api.point_op_dev(curve, "q4_add", p, p) == api.point_op_dev(curve, "dbl", p)).all()
    # non-trivial Z on both sides: 2 P + 3 Q style operands straight from the one-thread kernels
    a2 = api.point_op_dev(curve, "dbl", p)
    assert (api.point_op_dev(curve, "q4_dbl", a2) == api.point_op_dev(curve, "dbl", a2)).all()


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n,cbits,nb", [(1000, 8, 1), (4096, 10, 2), (1 << 14, 12, 6), (1 << 16, 14, 1), (1 << 16, 9, 3)])
def test_msm_with_q4_reduction_vs_oracle(curve, n, cbits, nb):
    """MSMs whose bucket set IS reduced by msm_reduce_q4_kernel / msm_window_sum_q4_kernel: a fixed-base table makes ONE bucket set per item
    (Ws = 1), and the quad-lane gate of msm.hip takes launches of at most 8 sets whose slices give >= 64 quads -- here 2^(cbits - 1) buckets
    in slices of one to a few buckets (the offset multiple meets its running sum: the P + P branch), for 1, 2, 3 and 6 items -- against
    cpu_ref.best_multiexp.  (without the table these sizes have 26 - 32 windows and never reached the quad kernels.)  The same
    cases run over the thread-per-slice kernels in test_thread_per_slice_reduction_in_a_subprocess."""
    bases = api.Bases.generate(curve, 5, 3, n)
    assert bases.precompute(cbits) == cbits
    assert api.get_option("reduce_q4") == (0 if os.environ.get("TRH_REDUCE_Q4_NESTED") else 1)
    sc = synth.field_elements(77 + n + nb, nb * n)
    got = bases.msm_batch_dev(api.DeviceBuffer.from_host(sc), n, nb)
    host_bases = bases.download()
    for i in range(nb):
        want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc[i * n:(i + 1) * n], host_bases, threads=cpu_ref.hardware_threads()))
        assert (np.asarray(got[i])[:8] == want).all(), i
    bases.destroy()


def test_thread_per_slice_reduction_in_a_subprocess():
    """TRH_REDUCE_Q4=0 (read once per process): the MSM parity tests over the one-thread-per-slice reduction kernels, which remain the path of
    every launch that is not a latency chain"""
    if os.environ.get("TRH_REDUCE_Q4_NESTED"):
        pytest.skip("already inside the TRH_REDUCE_Q4=0 run")
    env = dict(os.environ, TRH_REDUCE_Q4="0", TRH_REDUCE_Q4_NESTED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_q4.py"), "-q", "-x", "-m", "gpu",
                        "-k", "msm"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
