"""GPU parity of the frontier expansion (hpmvs_expand_batch) against the oracle's restatement of the
candidate loops of CellProcessor::extend / ::branch (reference src/hpmvs/CellProcessor.cpp:84-178,
210-262).  Tolerance: identical -- candidate construction is a handful of float32 operations in the
reference's order, the refinement is the bit-identical optimize() path, the gates are comparisons."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def refined_parents(tiny_scene, tiny_seeds, gpu_scene):
    from hpmvs_amd import api
    b = api.Batch.from_seeds(tiny_seeds)
    api.optimize_batch(gpu_scene, b)
    keep = np.nonzero(b.ok)[0]
    assert len(keep) >= 20
    par = api.Batch(b.center[keep], b.normal[keep], b.scale[keep], b.n_images[keep], b.images[keep])
    return par


def oracle_parents(par):
    from oracle import oracle as orc
    P = (orc.Patch * par.n)()
    for k in range(par.n):
        P[k].center[:] = par.center[k].tolist()
        P[k].normal[:] = par.normal[k].tolist()
        P[k].scale = float(par.scale[k])
        P[k].n_images = int(par.n_images[k])
        for j in range(par.n_images[k]):
            P[k].images[j] = int(par.images[k, j])
    return P


@pytest.mark.parametrize("mode", [0, 1])
def test_expand_matches_oracle(mode, tiny_scene, tiny_seeds, oracle_scene, gpu_scene):
    from hpmvs_amd import api
    from oracle import oracle as orc
    par = refined_parents(tiny_scene, tiny_seeds, gpu_scene)
    n, N = par.n, (6 if mode == 0 else 4)
    rng = np.random.default_rng(3 + mode)
    # octree leaves: extend keeps the candidate scale near the parent's (width = 2*scale/0.9); branch halves it.
    width = (par.scale * (2.0 / 0.9) * (1.0 if mode == 0 else 1.6)).astype(np.float32)
    # leaf centres near the parent centre, offset so that some branch candidates fall outside the leaf
    cc = (par.center[:, :3] + rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32) * width[:, None]).astype(np.float32)
    skip = (rng.uniform(size=n * N) < 0.15).astype(np.uint8) if mode == 0 else None
    out = api.expand_batch(gpu_scene, mode, par, cc, width, skip)
    ref = orc.expand_batch(oracle_scene, mode, oracle_parents(par), cc, width, skip, n_threads=8)
    assert out.n == n * N
    stage_cpu = np.array([ref[t].stage for t in range(n * N)])
    assert np.array_equal(stage_cpu, out.stage), (np.nonzero(stage_cpu != out.stage)[0][:10], stage_cpu[:24], out.stage[:24])
    assert np.array_equal(out.ok.astype(bool), stage_cpu == 0)
    # every candidate: constructed / refined centre, normal and scale identical
    for t in range(n * N):
        assert np.array_equal(np.array(ref[t].center[:], dtype=np.float32), out.center[t]), t
        assert np.array_equal(np.array(ref[t].normal[:], dtype=np.float32), out.normal[t]), t
        assert np.float32(ref[t].scale) == out.scale[t], t
        if stage_cpu[t] == 0:
            m = ref[t].n_images
            assert m == out.n_images[t] and list(ref[t].images[:m]) == list(out.images[t, :m]), t
    acc = int((stage_cpu == 0).sum())
    assert acc >= 10, f"only {acc} accepted candidates -- test scene too hard"
    if mode == 0:
        assert (stage_cpu[skip.astype(bool)] == 20).all()
    else:
        assert (stage_cpu == 20).sum() > 0 and (stage_cpu == 20).sum() < n * N  # the pre-gate is exercised both ways


def test_expand_rejects_bad_arguments(tiny_scene, tiny_seeds, gpu_scene):
    from hpmvs_amd import api
    par = refined_parents(tiny_scene, tiny_seeds, gpu_scene)
    with pytest.raises(api.HpmvsError):
        api.expand_batch(gpu_scene, 7, par, par.center[:, :3], par.scale)
    empty = api.Batch(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), np.zeros(0, np.float32),
                      np.zeros(0, np.int32), np.zeros((0, par.max_images), np.int32))
    out = api.expand_batch(gpu_scene, 0, empty, np.zeros((0, 3), np.float32), np.zeros(0, np.float32))
    assert out.n == 0


def test_expand_with_device_pointers_on_a_stream(tiny_scene, tiny_seeds, gpu_scene):
    """on_device = 1: every array is a device pointer and the call only enqueues on the given hipStream_t
    (torch is just the allocator / stream here).  Same results as the host-pointer form."""
    import ctypes as C
    import torch
    from hpmvs_amd import api
    par = refined_parents(tiny_scene, tiny_seeds, gpu_scene)
    n, N = par.n, 6
    width = (par.scale * np.float32(2.0 / 0.9)).astype(np.float32)
    cc = np.ascontiguousarray(par.center[:, :3])
    want = api.expand_batch(gpu_scene, api.EXPAND_EXTEND, par, cc, width)
    dev = torch.device("cuda", 0)

    def dbatch(m, src=None):
        t = dict(center=torch.zeros((m, 4), dtype=torch.float32, device=dev), normal=torch.zeros((m, 4), dtype=torch.float32, device=dev),
                 scale=torch.zeros(m, dtype=torch.float32, device=dev), n_images=torch.zeros(m, dtype=torch.int32, device=dev),
                 images=torch.full((m, par.max_images), -1, dtype=torch.int32, device=dev),
                 ok=torch.zeros(m, dtype=torch.uint8, device=dev), stage=torch.zeros(m, dtype=torch.int32, device=dev))
        if src is not None:
            for k in ("center", "normal", "scale", "n_images", "images"):
                t[k].copy_(torch.from_numpy(getattr(src, k)))
        b = api.PatchBatch()
        b.n, b.max_images = m, par.max_images
        for k, v in t.items():
            setattr(b, k, v.data_ptr())
        return t, b

    pt, pb = dbatch(n, par)
    ot, ob = dbatch(n * N)
    dcc, dcw = torch.from_numpy(cc).to(dev), torch.from_numpy(width).to(dev)
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))  # the uploads above ran on the current stream
    o = api.default_options()
    with torch.cuda.stream(stream):
        rc = api.lib().hpmvs_expand_batch(gpu_scene.h, C.byref(o), api.EXPAND_EXTEND, C.byref(pb), dcc.data_ptr(), dcw.data_ptr(),
                                          None, C.byref(ob), 1, C.c_void_p(stream.cuda_stream))
    assert rc == 0, api.lib().hpmvs_last_error()
    stream.synchronize()
    assert np.array_equal(ot["ok"].cpu().numpy(), want.ok)
    assert np.array_equal(ot["stage"].cpu().numpy(), want.stage)
    assert np.array_equal(ot["center"].cpu().numpy(), want.center)
    assert np.array_equal(ot["normal"].cpu().numpy(), want.normal)
    assert np.array_equal(ot["scale"].cpu().numpy(), want.scale)
    acc = want.ok.astype(bool)
    assert np.array_equal(ot["n_images"].cpu().numpy()[acc], want.n_images[acc])


def test_expand_and_concurrent_host_batches_while_callers_keep_an_open_batch_busy(tiny_scene, tiny_seeds, gpu_scene):
    """hpmvs_expand_batch takes part in the Exclusive protocol (its allocations and frees quiesce the open
    batch), and host-staged calls hold the scene lock only around their allocations and frees: an expandBatch and two
    big host batches issued from three threads at once, while 8 threads call optimize-of-one in a loop, all finish
    promptly with the results of undisturbed runs."""
    import threading, time
    from hpmvs_amd import api
    par = refined_parents(tiny_scene, tiny_seeds, gpu_scene)
    n = len(tiny_seeds.scale)
    width = (par.scale * np.float32(2.0 / 0.9)).astype(np.float32)
    cc = np.ascontiguousarray(par.center[:, :3])
    want_exp = api.expand_batch(gpu_scene, api.EXPAND_EXTEND, par, cc, width)
    ref = api.Batch.from_seeds(tiny_seeds, np.arange(n))
    api.optimize_batch(gpu_scene, ref)
    stop = threading.Event()
    errors, calls = [], [0] * 8

    def caller(k):
        i = 0
        try:
            while not stop.is_set():
                j = (k * 11 + i) % n
                b = api.Batch.from_seeds(tiny_seeds, np.array([j]))
                api.optimize_batch(gpu_scene, b)
                if b.ok[0] != ref.ok[j] or (b.ok[0] and not np.array_equal(b.center[0], ref.center[j])):
                    errors.append(("caller", k, j))
                i += 1
                calls[k] = i
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    results = {}

    def big(tag):
        try:
            b = api.Batch.from_seeds(tiny_seeds, np.tile(np.arange(n), 30))
            api.optimize_batch(gpu_scene, b)
            results[tag] = b
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def expand():
        try:
            results["exp"] = [api.expand_batch(gpu_scene, api.EXPAND_EXTEND, par, cc, width) for _ in range(3)]
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=caller, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    time.sleep(0.3)
    before = sum(calls)
    work = [threading.Thread(target=big, args=("a",)), threading.Thread(target=big, args=("b",)), threading.Thread(target=expand)]
    t0 = time.time()
    for t in work:
        t.start()
    for t in work:
        t.join(timeout=120)
    dt = time.time() - t0
    time.sleep(0.3)
    after = sum(calls)
    stop.set()
    for t in threads:
        t.join(timeout=60)
    assert not errors, errors[:3]
    assert dt < 30.0, dt
    assert before > 0 and after > before, (before, after)
    for tag in ("a", "b"):
        assert np.array_equal(results[tag].ok.reshape(30, n), np.tile(ref.ok, (30, 1)))
        assert np.array_equal(results[tag].center.reshape(30, n, 4)[17], ref.center)
    for e in results["exp"]:
        assert np.array_equal(e.ok, want_exp.ok) and np.array_equal(e.stage, want_exp.stage)
        assert np.array_equal(e.center, want_exp.center) and np.array_equal(e.normal, want_exp.normal)
