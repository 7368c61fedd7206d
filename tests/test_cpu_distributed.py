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
    with pytest.raises(ValueError):  # wider than the ABI's 256
        dd.pack_records(*base(1), t(np.array([1], np.int32)), t(np.zeros((1, 257), np.int32)))


# ---- bench.py's own multi-rank logic: Plan (workload default, counts, lo/hi), the slicing of the seed
# arrays, pack -> all-gather -> my_slice, the per-rank figures and the MAX over ranks -- executed at world size 2 on
# gloo by calling bench.main() itself with a backend whose kernel call is the CPU oracle.
class _OracleBackend:
    """bench.GpuBackend's interface on CPU tensors: refine = the oracle, pack = the torch packer, marks = wall clock."""
    name = "oracle-stub"
    dist_backend = "gloo"

    def __init__(self):
        self.device = torch.device("cpu")
        self.calls = 0

    def init_dist(self):
        dist.init_process_group("gloo")

    def make_scene(self, plan):
        from hpmvs_amd import synth
        from oracle import oracle as orc
        scene = synth.make_scene(plan.views, plan.W, plan.H, n_waves=24)
        self.osc = orc.OracleScene(scene)
        return scene

    def mark(self):
        import time
        return time.perf_counter()

    @staticmethod
    def ms_between(a, b):
        return 1e3 * (b - a)

    def refine(self, work, n, max_images):
        from oracle import oracle as orc
        self.calls += 1
        P = (orc.Patch * n)()
        cen, nor, sca = work["center"].numpy(), work["normal"].numpy(), work["scale"].numpy()
        nim, img = work["n_images"].numpy(), work["images"].numpy()
        for k in range(n):
            P[k].center[:] = cen[k].tolist(); P[k].normal[:] = nor[k].tolist(); P[k].scale = float(sca[k])
            P[k].n_images = int(nim[k])
            for j in range(int(nim[k])):
                P[k].images[j] = int(img[k, j])
        orc.optimize_batch(self.osc, P, n_threads=2)
        for k in range(n):
            okk = P[k].stage == 0
            work["ok"][k] = 1 if okk else 0
            work["stage"][k] = P[k].stage; work["nevals"][k] = P[k].nevals; work["ngrabs"][k] = P[k].ngrabs
            work["fmin"][k] = P[k].fmin
            if okk:
                work["center"][k] = torch.tensor(list(P[k].center)); work["normal"][k] = torch.tensor(list(P[k].normal))
                work["color"][k] = torch.tensor(list(P[k].color)); work["n_images"][k] = P[k].n_images
                work["images"][k, :P[k].n_images] = torch.tensor(list(P[k].images[:P[k].n_images]), dtype=torch.int32)

    def pack(self, work, n, max_images, records):
        from hpmvs_amd import distributed as dd
        records.copy_(dd.pack_records(work["center"], work["normal"], work["color"], work["scale"], work["fmin"], work["ok"],
                                      work["n_images"], work["images"], check=True))

    def sync(self):
        pass


def _bench_worker(rank, world, port, argv, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    if world == 1:
        os.environ["HPMVS_BENCH_FORCE_DIST"] = "1"
    import bench
    out = bench.main(argv, backend=_OracleBackend())
    g = out.pop("_gathered")
    q.put((rank, out, g.numpy().copy()))


def _run_bench(world, argv):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, out, g = q.get(timeout=600)
        got[r] = (out, g)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_bench_plan_defaults_and_shards():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.default_workload(1) == "c2_50v_4k_1e6" and bench.default_workload(8) == "c3_50v_4k_1e6_sharded"
    n = 0
    for r in range(8):
        p = bench.Plan(bench.default_workload(8), None, 0, r, 8)
        assert p.scaling == "strong" and p.n_job == 1_000_000 and p.n_total == 1_000_000 and p.lo == n
        assert p.counts == [125_000] * 8 and p.n_seeds == 125_000
        n = p.hi
    assert n == 1_000_000
    p1 = bench.Plan(bench.default_workload(1), None, 0, 0, 1)
    assert p1.scaling == "weak" and p1.n_seeds == 1_000_000 and p1.n_job == 1_000_000   # N = 1: the same 1e6 seeds
    pw = bench.Plan("c2_50v_4k_1e6", "weak", 0, 3, 4)
    assert pw.n_job == 4_000_000 and pw.n_seeds == 1_000_000 and pw.seed_value != bench.Plan("c2_50v_4k_1e6", "weak", 0, 2, 4).seed_value
    pr = bench.Plan("c0_tiny", "strong", 31, 1, 2)   # ragged
    assert (pr.lo, pr.hi, pr.counts) == (16, 31, [16, 15])


def test_bench_main_two_ranks_on_gloo_equals_one_rank():
    """bench.main() itself at world size 2 (strong scaling of a 31-seed tiny round, ragged shards): every rank ends with
    the same full round, equal to the 1-rank round byte for byte; rank 0's line carries per-rank figures and the job's
    time is the maximum over the ranks."""
    from hpmvs_amd import distributed as dd
    argv = ["--gpus", "2", "--workload", "c0_tiny", "--scaling", "strong", "--patches", "31", "--steps", "2", "--warmup", "1",
            "--no-cpu", "--check-gather"]
    two = _run_bench(2, argv)
    one = _run_bench(1, ["--gpus", "1"] + argv[2:])
    out0, g0 = two[0]
    out1, g1 = two[1]
    assert g0.shape == (31, dd.RECORD_BYTES) and np.array_equal(g0, g1)
    assert np.array_equal(g0, one[0][1])
    assert out0["n_gpus"] == 2 and out0["scaling"] == "strong" and out0["config"]["workload"] == "c0_tiny"
    assert out0["config"]["seeds_per_gpu"] == 16 and out0["config"]["seeds_per_step_all_gpus"] == 31
    assert out1["config"]["seeds_per_gpu"] == 15
    pr = out0["per_rank"]
    assert pr["patches"] == [16, 15] and len(pr["kernel_ms"]) == 2 and all(v > 0 for v in pr["kernel_ms"])
    assert all(v >= 0 for v in pr["pack_plus_allgather_ms"])
    # the job's step time is the slowest rank's, and `value` is the whole round over it
    assert abs(out0["ms_per_step"] - max(pr["ms_per_step"])) < 1e-6 and out0["ms_per_step"] == out1["ms_per_step"]
    assert abs(out0["value"] - 31 * 2 / (out0["ms_per_step"] * 2e-3)) < 1e-6 * out0["value"]
    gc = out0["gather_check"]
    assert gc["records"] == 16 and gc["gathered_total"] == 31 and gc["bytes_equal_python_packer"] and gc["fields_equal_results"]
    u = dd.unpack_records(torch.from_numpy(g0))
    assert u["ok"].sum() >= 10


def test_bench_main_eight_ranks_on_gloo_ragged_shards_in_tile_order():
    """bench.main() at world size 8 (what the driver's SCALE run launches), strong scaling of a 61-seed round: shards of
    8, 8, 8, 8, 8, 7, 7, 7 seeds, each rank's shard bucket-sorted by (reference image, tile) before it is refined
    (`--order tiles`).  Every rank ends with the same gathered round; it holds exactly the records of the
    one-rank round (the tile order permutes them inside a shard, so the comparison is on the sorted records); the line's
    `per_rank.collective` block checks itself (8 ranks seen, the shards add up to the job)."""
    from hpmvs_amd import distributed as dd
    argv = ["--gpus", "8", "--workload", "c0_tiny", "--scaling", "strong", "--patches", "61", "--steps", "1", "--warmup", "0",
            "--no-cpu", "--check-gather", "--order", "tiles"]
    eight = _run_bench(8, argv)
    one = _run_bench(1, ["--gpus", "1"] + argv[2:-2])     # the one-rank round in the seeds' own order
    g0 = eight[0][1]
    assert g0.shape == (61, dd.RECORD_BYTES)
    for r in range(1, 8):
        assert np.array_equal(eight[r][1], g0), f"rank {r} gathered a different round"
    as_rows = lambda g: sorted(bytes(row) for row in g)
    assert as_rows(g0) == as_rows(one[0][1])
    out0 = eight[0][0]
    pr = out0["per_rank"]
    assert pr["patches"] == [8, 8, 8, 8, 8, 7, 7, 7] and len(pr["kernel_ms"]) == 8
    co = pr["collective"]
    assert co["backend"] == "gloo" and co["world_size_seen"] == 8 and co["ranks_seen"] == list(range(8))
    assert co["gathered_records"] == 61 and co["gathered_equals_job"] and co["shards_sum_to_job"]
    assert out0["n_gpus"] == 8 and out0["config"]["order"] == "tiles" and out0["config"]["seeds_per_step_all_gpus"] == 61
    assert all(abs(eight[r][0]["ms_per_step"] - out0["ms_per_step"]) < 1e-9 for r in range(8))   # MAX over ranks, on every rank


# ---- lists longer than the record's 64 ids: tails (round 6; include/hpmvs_amd.h: hpmvs_record_tail)
def _wide_round(n, seed):
    rng = np.random.default_rng(seed)
    nim = rng.integers(3, 40, n).astype(np.int32)
    long = rng.random(n) < 0.2
    nim[long] = rng.integers(65, 201, int(long.sum()))
    ok = (rng.random(n) < 0.8).astype(np.uint8)
    images = np.full((n, 200), -1, np.int32)
    for k in range(n):
        images[k, :nim[k]] = rng.permutation(300)[:nim[k]]
    f = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    return dict(center=f(n, 4), normal=f(n, 4), color=f(n, 3), scale=f(n), fmin=rng.standard_normal(n), ok=ok, n_images=nim, images=images)


def test_tails_carry_the_ids_beyond_64_through_pack_and_unpack():
    from hpmvs_amd import distributed as dd
    R = _wide_round(57, 5)
    t = torch.from_numpy
    rec = dd.pack_records(t(R["center"]), t(R["normal"]), t(R["color"]), t(R["scale"]), t(R["fmin"]), t(R["ok"]), t(R["n_images"]), t(R["images"]))
    tails = dd.pack_tails(t(R["ok"]), t(R["n_images"]), t(R["images"]))
    want = np.nonzero((R["ok"] != 0) & (R["n_images"] > 64))[0]
    assert rec.shape == (57, dd.RECORD_BYTES) and tails.shape == (len(want), dd.TAIL_BYTES) and len(want) >= 3
    u = dd.unpack_records(rec, tails)
    assert np.array_equal(u["n_images"], R["n_images"])
    for k in range(57):
        m = R["n_images"][k]
        if R["ok"][k] or m <= 64:
            assert list(u["images"][k, :m]) == list(R["images"][k, :m]), k
        else:   # not refined: no tail travels for it (its record says ok = 0)
            assert list(u["images"][k, :64]) == list(R["images"][k, :64])
    # no long list anywhere: the exchange is exactly rounds 1-5's (no tails, 64 columns)
    S = _wide_round(20, 6); S["n_images"] = np.minimum(S["n_images"], 30)
    assert dd.pack_tails(t(S["ok"]), t(S["n_images"]), t(S["images"])).shape[0] == 0


def _tails_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from hpmvs_amd import distributed as dd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = _wide_round(41, 9)   # the same round on every rank
    counts = [dd.shard_range(41, r, world)[1] - dd.shard_range(41, r, world)[0] for r in range(world)]
    lo, hi = dd.shard_range(41, rank, world)
    t = lambda a: torch.from_numpy(a[lo:hi].copy())
    rec = dd.pack_records(t(R["center"]), t(R["normal"]), t(R["color"]), t(R["scale"]), t(R["fmin"]), t(R["ok"]), t(R["n_images"]), t(R["images"]))
    tails = dd.pack_tails(t(R["ok"]), t(R["n_images"]), t(R["images"]))
    g, gt = dd.all_gather_round(rec, tails, counts)
    u = dd.unpack_records(g, gt)
    q.put((rank, g.numpy().copy(), gt.numpy().copy(), u["images"].copy(), u["n_images"].copy()))
    dist.destroy_process_group()


def test_three_rank_round_with_tails_equals_the_one_rank_round():
    """Ragged shards of a round in which a fifth of the lists is longer than 64 ids: every rank ends with the same records and
    the same tails (patch indices rebased to the round), and unpacking gives back every refined list in full."""
    from hpmvs_amd import distributed as dd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33100 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_tails_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(3):
        r, g, gt, img, nim = q.get(timeout=300)
        got[r] = (g, gt, img, nim)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    R = _wide_round(41, 9)
    t = torch.from_numpy
    one = dd.pack_records(t(R["center"]), t(R["normal"]), t(R["color"]), t(R["scale"]), t(R["fmin"]), t(R["ok"]), t(R["n_images"]), t(R["images"])).numpy()
    one_t = dd.pack_tails(t(R["ok"]), t(R["n_images"]), t(R["images"])).numpy()
    for r in range(3):
        g, gt, img, nim = got[r]
        assert np.array_equal(g, one) and np.array_equal(gt, one_t), r
        for k in range(41):
            if R["ok"][k]:
                assert list(img[k, :R["n_images"][k]]) == list(R["images"][k, :R["n_images"][k]]), (r, k)
