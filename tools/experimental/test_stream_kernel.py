"""kernel_stream.hip (HPMVS_KERNEL=stream: patches queue per optimiser block in HBM records, block visits page the
state into LDS, sc1 hand-offs between wavefronts) must produce the bytes of the wavefront-bound kernel: the blocks
are the same functions, only which wavefront runs which block for which patch changes.  Tolerance: none."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("ok", "center", "normal", "n_images", "images", "color", "ncc", "fmin", "x", "result", "nevals", "stage", "ngrabs")


def _both(gpu, seeds, idx=None):
    from hpmvs_amd import api
    out = {}
    old = os.environ.get("HPMVS_KERNEL")
    try:
        for mode in ("wave", "stream"):
            os.environ["HPMVS_KERNEL"] = mode
            b = api.Batch.from_seeds(seeds, idx)
            api.optimize_batch(gpu, b)
            out[mode] = b
    finally:
        if old is None:
            os.environ.pop("HPMVS_KERNEL", None)
        else:
            os.environ["HPMVS_KERNEL"] = old
    return out["wave"], out["stream"]


def test_stream_kernel_equals_wavefront_kernel_tiny(tiny_seeds, gpu_scene):
    w, s = _both(gpu_scene, tiny_seeds)
    for f in FIELDS:
        assert np.array_equal(getattr(w, f), getattr(s, f)), f
    # ragged / degenerate batches: one patch, patches that fail before the optimiser starts
    w, s = _both(gpu_scene, tiny_seeds, np.arange(1))
    for f in FIELDS:
        assert np.array_equal(getattr(w, f), getattr(s, f)), f


def test_stream_kernel_equals_wavefront_kernel_mid_size():
    """40 000 seeds on 12 x 1080p: more patches than in-flight records per shard are not needed to recycle records
    (131 072 records), but every queue, the record recycling through FETCH and all block visits are exercised; the
    oracle parity of the wavefront-bound kernel (test_gpu_big_configs.py) then carries over byte for byte."""
    import torch
    from hpmvs_amd import api, synth
    scene = synth.make_scene(12, 1920, 1080, n_waves=24, device=torch.device("cuda", 0))
    gpu = api.Scene(scene, device=0)
    seeds = synth.make_seeds(scene, 40_000, start_level=4, max_images=32)
    w, s = _both(gpu, seeds)
    assert w.ok.sum() > 10_000
    for f in FIELDS:
        assert np.array_equal(getattr(w, f), getattr(s, f)), f
    # few records in flight: every record is recycled many times
    os.environ["HPMVS_STREAM_SLOTS"] = "4096"
    try:
        w2, s2 = _both(gpu, seeds, np.arange(20_000))
    finally:
        os.environ.pop("HPMVS_STREAM_SLOTS", None)
    for f in FIELDS:
        assert np.array_equal(getattr(w2, f), getattr(s2, f)), f
    gpu.close()
