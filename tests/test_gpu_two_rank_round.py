"""One expansion round of BASELINE configs[3] with the REAL kernel at world size 2 (VERDICT r4 #1b).

tests/test_cpu_distributed.py runs the 2-rank round with the kernel stubbed by the oracle; tests/test_gpu_bench_dist.py
runs the real kernel at one rank.  Here two processes share the one GPU of the box (RCCL refuses two ranks on one
device, so the collective goes over gloo with the records staged through host memory -- the shard / pack / gather /
unpack path is otherwise the product's): every rank

    replicates the 50 x 3840x2160 scene, takes `shard_range(n, rank, 2)` of the SAME 2e5 seeds (reference
    main.cpp:145-181: one round over one patch set; Scene.cpp:114-178: the seed loop that is cut),
    refines it with hpmvs_optimize_batch on device arrays, fills its slice of the exchange buffer with
    hpmvs_pack_records (the product's kernel), all-gathers ONCE, and unpacks the gathered round with
    hpmvs_unpack_records into a full-size device batch.

The gathered round of EVERY rank must equal the 1-rank round byte for byte, and the unpacked fields must equal the 1-rank
run's un-gathered results.  n is odd so that the shards are ragged (padded all-gather)."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 200_001
MAX_IMAGES = 32


def _worker(rank, world, port, n, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from hpmvs_amd import api, synth
    from hpmvs_amd import distributed as dd
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    L = api.lib()
    scene = synth.make_scene(50, 3840, 2160, n_waves=24, device=dev)
    gpu = api.Scene(scene, device=0)
    seeds = synth.make_seeds(scene, n, start_level=4, max_images=MAX_IMAGES)
    lo, hi = dd.shard_range(n, rank, world)
    counts = [dd.shard_range(n, r, world)[1] - dd.shard_range(n, r, world)[0] for r in range(world)]
    mine = synth.Seeds(center=seeds.center[lo:hi], normal=seeds.normal[lo:hi], scale=seeds.scale[lo:hi],
                       n_images=seeds.n_images[lo:hi], images=seeds.images[lo:hi], truth=seeds.truth[lo:hi])
    work = bench.device_batch(mine, dev, MAX_IMAGES)
    m = hi - lo
    b = bench.c_batch(work, m, MAX_IMAGES)
    opts = api.default_options()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.hpmvs_optimize_batch(gpu.h, C.byref(opts), C.byref(b), 1, st) == 0, L.hpmvs_last_error()
    rec = torch.empty((m, dd.RECORD_BYTES), dtype=torch.uint8, device=dev)
    assert L.hpmvs_pack_records(C.byref(b), C.c_void_p(rec.data_ptr()), st) == 0, L.hpmvs_last_error()
    torch.cuda.synchronize()
    full = dd.all_gather_records(rec.cpu(), counts)          # ONE collective per round (gloo: host tensors)
    assert full.shape == (n, dd.RECORD_BYTES)
    # the gathered round back into a full-size SoA batch on the device, by the product's unpack kernel
    full_dev = full.to(dev)
    out = bench.device_batch(synth.Seeds(center=np.zeros((n, 4), np.float32), normal=np.zeros((n, 4), np.float32),
                                         scale=np.zeros(n, np.float32), n_images=np.zeros(n, np.int32),
                                         images=np.full((n, MAX_IMAGES), -7, np.int32), truth=np.zeros((n, 3))), dev, MAX_IMAGES)
    ob = bench.c_batch(out, n, MAX_IMAGES)
    assert L.hpmvs_unpack_records(C.c_void_p(full_dev.data_ptr()), n, C.byref(ob), st) == 0, L.hpmvs_last_error()
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"gathered_w{world}_r{rank}.npy"), full.numpy())
    np.savez(os.path.join(outdir, f"fields_w{world}_r{rank}.npz"),
             **{k: out[k].cpu().numpy() for k in ("center", "normal", "color", "scale", "fmin", "ok", "n_images", "images")})
    if world == 1:  # the un-gathered results of the one-rank round
        np.savez(os.path.join(outdir, "direct_w1.npz"),
                 **{k: work[k].cpu().numpy() for k in ("center", "normal", "color", "scale", "fmin", "ok", "n_images", "images", "nevals", "stage")})
    gpu.close()
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n, outdir):
    ctx = mp.get_context("spawn")
    port = 29700 + (os.getpid() % 1500) + 7 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, outdir)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"


def test_two_ranks_with_the_real_kernel_equal_the_one_rank_round():
    from hpmvs_amd import api
    if api.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests need the MI355X box (no CPU fallback exists)")
    with tempfile.TemporaryDirectory() as d:
        _run(1, N, d)
        _run(2, N, d)
        one = np.load(os.path.join(d, "gathered_w1_r0.npy"))
        assert one.shape == (N, 192)
        direct = np.load(os.path.join(d, "direct_w1.npz"))
        ok = direct["ok"].astype(bool)
        assert 0.5 * N < ok.sum() < N
        for r in range(2):
            two = np.load(os.path.join(d, f"gathered_w2_r{r}.npy"))
            assert np.array_equal(two, one), f"rank {r}: the gathered round differs from the one-rank round"
            f = np.load(os.path.join(d, f"fields_w2_r{r}.npz"))
            # what every rank holds after the round == what one GPU computes without any exchange
            assert np.array_equal(f["ok"], direct["ok"])
            for k in ("center", "normal", "scale", "fmin", "n_images"):
                assert np.array_equal(f[k], direct[k]), (r, k)
            assert np.array_equal(f["color"][ok], direct["color"][ok])
            live = np.arange(MAX_IMAGES)[None, :] < np.clip(direct["n_images"], 0, None)[:, None]
            assert np.array_equal(f["images"][live], direct["images"][live]) and np.all(f["images"][~live] == -1)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "two_rank_round.json"), "w") as fh:
            import json
            fh.write(json.dumps(dict(n=N, world=2, collective="gloo all_gather_into_tensor (2 processes on one GPU)",
                                     refined=int(ok.sum()), gathered_equal_one_rank=True, unpacked_equal_direct=True)) + "\n")
