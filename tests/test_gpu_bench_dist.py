"""bench.py's multi-GPU leg on ONE GPU: HPMVS_BENCH_FORCE_DIST=1 makes the single rank initialise RCCL and run the
round exactly as the N-rank job does -- refine the rank's shard (configs[3]: the c2 batch cut with shard_range),
hpmvs_pack_records (the product's packing kernel), ONE all-gather of the 192-byte records -- and --check-gather
compares the gathered records with the un-gathered results byte for byte."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_dist_round_on_one_gpu():
    env = dict(os.environ, HPMVS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3_50v_4k_1e6_sharded", "--gpus", "1",
                        "--steps", "2", "--warmup", "1", "--no-cpu", "--patches", "200000", "--check-gather"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["value"] > 0
    g = line["gather_check"]
    assert g["collective"] == "all_gather_into_tensor over RCCL (1 rank)" and g["records"] == 200000
    assert g["bytes_equal_python_packer"] and g["fields_equal_results"], g
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_c3_forced_dist_n1.json"), "w") as fh:
        fh.write(json.dumps(line) + "\n")


def test_c_and_python_record_codecs_agree_including_rejection_codes():
    """the 192-byte record has two implementations -- the C ABI's kernels (hpmvs_pack_records /
    hpmvs_unpack_records) and hpmvs_amd.distributed.  Packed by one and unpacked by the other, in both directions,
    with a NEGATIVE image count (seed_init's rejection codes travel as -k), an empty list and a full 64-id list."""
    import ctypes as C
    import torch
    from hpmvs_amd import api
    from hpmvs_amd import distributed as dd
    L = api.lib()
    dev = torch.device("cuda", 0)
    n, m = 6, 64
    rng = np.random.default_rng(7)
    nim = np.array([-3, 64, 0, 5, -11, 1], np.int32)
    imgs = np.full((n, m), -1, np.int32)
    for k in range(n):
        if nim[k] > 0:
            imgs[k, :nim[k]] = rng.integers(0, 65535, nim[k])
    host = dict(center=rng.random((n, 4), np.float32), normal=rng.random((n, 4), np.float32), scale=rng.random(n, np.float32),
                n_images=nim, images=imgs, ok=np.array([0, 1, 0, 1, 0, 1], np.uint8), color=rng.random((n, 3), np.float32),
                ncc=np.zeros(n, np.float32), fmin=rng.random(n), x=np.zeros((n, 3)), result=np.zeros(n, np.int32),
                nevals=np.zeros(n, np.int32), stage=np.zeros(n, np.int32), ngrabs=np.zeros(n, np.int32))
    t = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}

    def c_batch(tt):
        b = api.PatchBatch()
        b.n, b.max_images = n, m
        for k in host:
            setattr(b, k, tt[k].data_ptr())
        return b

    # C pack -> Python unpack
    rec = torch.empty((n, dd.RECORD_BYTES), dtype=torch.uint8, device=dev)
    b = c_batch(t)
    assert L.hpmvs_pack_records(C.byref(b), C.c_void_p(rec.data_ptr()), None) == 0, L.hpmvs_last_error()
    torch.cuda.synchronize()
    u = dd.unpack_records(rec.cpu())
    assert np.array_equal(u["n_images"], nim) and np.array_equal(u["ok"], host["ok"].astype(bool))
    live = np.arange(m)[None, :] < np.clip(nim, 0, None)[:, None]
    assert np.array_equal(u["images"][live], imgs[live]) and np.all(u["images"][~live] == -1)
    assert np.array_equal(u["center"], host["center"]) and np.array_equal(u["fmin"], host["fmin"])
    # the Python packer writes the same bytes
    tp = torch.from_numpy
    ref = dd.pack_records(tp(host["center"]), tp(host["normal"]), tp(host["color"]), tp(host["scale"]), tp(host["fmin"]),
                          tp(host["ok"]), tp(nim), tp(imgs))
    assert torch.equal(ref, rec.cpu())
    # Python pack -> C unpack
    out = {k: torch.zeros_like(v) for k, v in t.items()}
    out["images"].fill_(-7)
    ob = c_batch(out)
    rec2 = ref.to(dev)
    assert L.hpmvs_unpack_records(C.c_void_p(rec2.data_ptr()), n, C.byref(ob), None) == 0, L.hpmvs_last_error()
    torch.cuda.synchronize()
    assert np.array_equal(out["n_images"].cpu().numpy(), nim)
    assert np.array_equal(out["ok"].cpu().numpy(), host["ok"])
    got = out["images"].cpu().numpy()
    assert np.array_equal(got[live], imgs[live]) and np.all(got[~live] == -1)
    assert np.array_equal(out["center"].cpu().numpy(), host["center"])


def test_c_and_python_tail_codecs_agree_for_lists_longer_than_64_ids():
    """Round 6: rows of up to HPMVS_MAX_IMAGES = 256 ids.  The record keeps the first 64, the rest travels as a 392-byte tail
    (hpmvs_pack_record_tails / hpmvs_unpack_record_tails = hpmvs_amd.distributed.pack_tails / unpack_records): same bytes from
    both packers, tails in patch order and only for refined patches, and unpacking restores every refined list in full."""
    import ctypes as C
    import torch
    from hpmvs_amd import api
    from hpmvs_amd import distributed as dd
    L = api.lib()
    dev = torch.device("cuda", 0)
    n, m = 700, 200
    rng = np.random.default_rng(11)
    nim = rng.integers(3, 50, n).astype(np.int32)
    long = rng.random(n) < 0.15
    nim[long] = rng.integers(65, m + 1, int(long.sum()))
    imgs = np.full((n, m), -1, np.int32)
    for k in range(n):
        imgs[k, :nim[k]] = rng.permutation(400)[:nim[k]]
    host = dict(center=rng.random((n, 4), np.float32), normal=rng.random((n, 4), np.float32), scale=rng.random(n, np.float32),
                n_images=nim, images=imgs, ok=(rng.random(n) < 0.8).astype(np.uint8), color=rng.random((n, 3), np.float32),
                ncc=np.zeros(n, np.float32), fmin=rng.random(n), x=np.zeros((n, 3)), result=np.zeros(n, np.int32),
                nevals=np.zeros(n, np.int32), stage=np.zeros(n, np.int32), ngrabs=np.zeros(n, np.int32))
    t = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}

    def c_batch(tt):
        b = api.PatchBatch()
        b.n, b.max_images = n, m
        for k in host:
            setattr(b, k, tt[k].data_ptr())
        return b

    tp = torch.from_numpy
    want_rec = dd.pack_records(tp(host["center"]), tp(host["normal"]), tp(host["color"]), tp(host["scale"]), tp(host["fmin"]),
                               tp(host["ok"]), tp(nim), tp(imgs))
    want_tails = dd.pack_tails(tp(host["ok"]), tp(nim), tp(imgs))
    assert want_tails.shape[0] >= 20
    rec = torch.empty((n, dd.RECORD_BYTES), dtype=torch.uint8, device=dev)
    tails = torch.zeros((n, dd.TAIL_BYTES), dtype=torch.uint8, device=dev)
    b = c_batch(t)
    nt = C.c_int32(-1)
    L.hpmvs_pack_record_tails.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
    L.hpmvs_unpack_record_tails.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    assert L.hpmvs_pack_records(C.byref(b), C.c_void_p(rec.data_ptr()), None) == 0, L.hpmvs_last_error()
    assert L.hpmvs_pack_record_tails(C.byref(b), C.c_void_p(tails.data_ptr()), n, C.byref(nt), None) == 0, L.hpmvs_last_error()
    torch.cuda.synchronize()
    assert nt.value == want_tails.shape[0]
    assert torch.equal(rec.cpu(), want_rec) and torch.equal(tails.cpu()[:nt.value], want_tails)
    assert L.hpmvs_pack_record_tails(C.byref(b), C.c_void_p(tails.data_ptr()), 3, C.byref(nt), None) == -2   # too small a segment: refused
    # unpack on the device: records, then the tails (as if this shard sat at offset 0 of the round)
    out = {k: torch.zeros_like(v) for k, v in t.items()}
    out["images"].fill_(-7)
    ob = c_batch(out)
    assert L.hpmvs_unpack_records(C.c_void_p(rec.data_ptr()), n, C.byref(ob), None) == 0, L.hpmvs_last_error()
    assert L.hpmvs_unpack_record_tails(C.c_void_p(tails.data_ptr()), want_tails.shape[0], 0, C.byref(ob), None) == 0, L.hpmvs_last_error()
    torch.cuda.synchronize()
    got = out["images"].cpu().numpy()
    assert np.array_equal(out["n_images"].cpu().numpy(), nim)
    for k in range(n):
        upto = nim[k] if host["ok"][k] else min(nim[k], 64)   # (a patch that was not refined sends no tail)
        assert np.array_equal(got[k, :upto], imgs[k, :upto]), k
    u = dd.unpack_records(rec.cpu(), tails.cpu()[:want_tails.shape[0]])
    for k in range(n):
        if host["ok"][k]:
            assert list(u["images"][k, :nim[k]]) == list(imgs[k, :nim[k]]), k
