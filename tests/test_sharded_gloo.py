"""CPU test of the N>1 path: world_size-2 gloo processes run the range-sharded MSM driver
(tiny_ram_halo2_amd/sharded.py).  There is no GPU here, so each rank's local MSM is computed by the
oracle (allowed in tests); the sharding, the all-gather and the host-side combine
(`trh_point_sum` in libtrh.so) are the product code under test."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, curve, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import cpu_ref
    from tiny_ram_halo2_amd import sharded, synth

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = sharded.shard_range(n, rank, world)
    sc = synth.field_elements(0x5EED, hi - lo, start=lo)
    bases = cpu_ref.gen_bases(curve, synth.BASE_S0 + lo * synth.BASE_D, synth.BASE_D, hi - lo, threads=2)

    def local():
        j = cpu_ref.best_multiexp(curve, sc, bases, threads=2)
        a = cpu_ref.to_affine(curve, j)
        one = np.array([0, 0, 0, 0], np.uint64) if not a.any() else None
        # normalised Jacobian (Z = 1) like the device path returns
        import pasta as o
        z = np.array(o.CURVES[curve].base.limbs(1), np.uint64) if a.any() else np.zeros(4, np.uint64)
        return np.concatenate([a, z])

    res = sharded.sharded_msm(curve, local)
    np.save(os.path.join(out_dir, f"res{rank}.npy"), res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 1001])
def test_sharded_msm_world2(tmp_path, n):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ref
    from tiny_ram_halo2_amd import synth

    world, curve = 2, "pallas"
    port = 29500 + (os.getpid() % 2000) + n % 7
    mp.spawn(_worker, args=(world, port, n, curve, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "res0.npy")
    r1 = np.load(tmp_path / "res1.npy")
    assert (r0 == r1).all()
    sc = synth.field_elements(0x5EED, n)
    bases = cpu_ref.gen_bases(curve, synth.BASE_S0, synth.BASE_D, n, threads=2)
    want = cpu_ref.to_affine(curve, cpu_ref.best_multiexp(curve, sc, bases, threads=4))
    assert (r0[:8] == want).all()


def test_shard_range_covers_everything():
    from tiny_ram_halo2_amd import sharded
    for n in (0, 1, 7, 8, 1000, (1 << 26) + 3):
        for world in (1, 2, 3, 4, 8):
            spans = [sharded.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _col_worker(rank, world, port, ncol, out_dir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tiny_ram_halo2_amd import sharded

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    calls = []

    def local(lo, hi):   # stands for commit_lagrange_batch over columns [lo, hi): a recognisable 12-word result per column
        calls.append((lo, hi))
        return np.array([[c * 1000 + w for w in range(12)] for c in range(lo, hi)], dtype=np.uint64).reshape(hi - lo, 12)

    res = sharded.sharded_columns(ncol, local)
    assert calls == [sharded.shard_range(ncol, rank, world)]
    np.save(os.path.join(out_dir, f"cols{rank}.npy"), res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ncol", [1, 5, 64])
def test_sharded_columns_world2(tmp_path, ncol):
    """column-sharded commitments: every rank ends with all results in column order, ragged split included"""
    world = 2
    port = 31500 + (os.getpid() % 2000) + ncol % 11
    mp.spawn(_col_worker, args=(world, port, ncol, str(tmp_path)), nprocs=world, join=True)
    want = np.array([[c * 1000 + w for w in range(12)] for c in range(ncol)], dtype=np.uint64)
    for r in range(world):
        assert (np.load(tmp_path / f"cols{r}.npy") == want).all()
