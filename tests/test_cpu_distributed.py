"""world_size-2 gloo test of the multi-GPU path's host logic: contiguous patch-index sharding, the
128-byte record pack/unpack, and ONE all-gather per round (hpmvs_amd/distributed.py).  The per-shard
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
