"""world_size-2 gloo test of the multi-GPU path's host logic: contiguous patch-index sharding, the
192-byte record pack/unpack, and ONE all-gather per round (hpmvs_amd/distributed.py).  The per-shard
compute is done by the CPU oracle here (the GPU kernel is covered by the -m gpu tests)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import golden_scene
    from hpmvs_amd import distributed as dd
    from oracle import oracle as orc
    scene, seeds, _ = golden_scene()
    osc = orc.OracleScene(scene)
    lo, hi = dd.shard_range(n, rank, world)
    P = orc.patches_from_seeds(seeds, list(range(lo, hi)))
    orc.optimize_batch(osc, P, n_threads=1)
    m = hi - lo
    t = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt))
    imgs = np.full((m, 32), -1, np.int32)
    for k in range(m):
        imgs[k, :P[k].n_images] = P[k].images[:P[k].n_images]
    rec = dd.pack_records(t([list(p.center) for p in P], np.float32), t([list(p.normal) for p in P], np.float32),
                          t([list(p.color) for p in P], np.float32), t([p.scale for p in P], np.float32),
                          t([p.fmin for p in P], np.float64), t([p.stage == 0 for p in P], np.uint8),
                          t([p.n_images for p in P], np.int32), t(imgs, np.int32))
    counts = [dd.shard_range(n, r, world)[1] - dd.shard_range(n, r, world)[0] for r in range(world)]
    full = dd.all_gather_records(rec, counts)
    if rank == 0:
        q.put(full.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_shard_range_is_a_contiguous_partition():
    from hpmvs_amd import distributed as dd
    for n in (0, 1, 7, 64, 100001):
        for world in (1, 2, 3, 8):
            r = [dd.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_round_equals_single_rank_round():
    from hpmvs_amd import distributed as dd
    n = 31  # odd: exercises the ragged (padded) all-gather
    two = _run(2, n)
    one = _run(1, n)
    assert two.shape == (n, dd.RECORD_BYTES)
    assert np.array_equal(two, one)
    u = dd.unpack_records(torch.from_numpy(two))
    assert u["ok"].sum() >= 10 and np.all(u["center"][:, 3][u["ok"]] == 1.0)


def test_records_carry_all_64_ids_and_16_bit_view_numbers():
    """no silent truncation to 32 ids / int16.  A 64-image patch and ids up to 65534 survive the
    round trip; unused slots come back as -1, never as view 0."""
    from hpmvs_amd import distributed as dd
    n = 5
    rng = np.random.default_rng(3)
    nim = np.array([64, 33, 3, 0, 40], np.int32)
    imgs = np.full((n, 64), -1, np.int32)
    for k in range(n):
        imgs[k, :nim[k]] = rng.integers(0, 65535, nim[k])
    imgs[0, 0] = 65534
    imgs[1, 5] = 40000
    t = torch.from_numpy
    rec = dd.pack_records(t(rng.random((n, 4), np.float32)), t(rng.random((n, 4), np.float32)),
                          t(rng.random((n, 3), np.float32)), t(rng.random(n, np.float32)),
                          t(rng.random(n)), t(np.ones(n, np.uint8)), t(nim), t(imgs))
    assert rec.shape == (n, dd.RECORD_BYTES) and dd.RECORD_BYTES == 192
    u = dd.unpack_records(rec)
    assert np.array_equal(u["n_images"], nim)
    assert np.array_equal(u["images"], imgs)


def test_records_refuse_what_they_cannot_carry():
    from hpmvs_amd import distributed as dd
    import pytest
    t = torch.from_numpy
    base = lambda n: (t(np.zeros((n, 4), np.float32)), t(np.zeros((n, 4), np.float32)), t(np.zeros((n, 3), np.float32)),
                      t(np.zeros(n, np.float32)), t(np.zeros(n)), t(np.ones(n, np.uint8)))
    with pytest.raises(ValueError):  # more images than id columns handed over
        dd.pack_records(*base(1), t(np.array([40], np.int32)), t(np.zeros((1, 32), np.int32)))
    with pytest.raises(ValueError):  # an id that does not fit 16 bits
        dd.pack_records(*base(1), t(np.array([2], np.int32)), t(np.array([[1, 70000] + [0] * 30], np.int32)))
    with pytest.raises(ValueError):  # 0xFFFF is the "no image" marker
        dd.pack_records(*base(1), t(np.array([1], np.int32)), t(np.array([[65535] + [0] * 31], np.int32)))
    with pytest.raises(ValueError):  # wider than the ABI's 64
        dd.pack_records(*base(1), t(np.array([1], np.int32)), t(np.zeros((1, 65), np.int32)))
