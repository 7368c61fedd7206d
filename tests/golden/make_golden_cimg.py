#!/usr/bin/env python3
"""Generates tests/golden/g4_cimg_pyramid.npz IN THE BUILD CONTAINER with the reference's GENUINE vendored CImg
(oracle/_ref/libref_cimg.so = /root/reference/thirdLibs/cimg/CImg.h behind the calls of Image.cpp:55-63).
Data only: input images (random bytes and a smooth ramp, even and odd sizes) and the pyramid levels CImg
produced for them.  Pins the oracle's orc_half_resize and the HIP half_resize_kernel where oracle/_ref is absent.

    python tests/golden/make_golden_cimg.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def inputs():
    rng = np.random.default_rng(0x43496D67)
    out = {}
    for name, (h, w) in dict(a=(96, 128), b=(61, 97), c=(33, 2), d=(7, 5)).items():
        out[name] = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:75, 0:101]
    out["ramp"] = np.stack([(xx * 255 // 100), (yy * 255 // 74), ((xx + yy) % 256)], axis=-1).astype(np.uint8)
    sat = rng.integers(0, 2, size=(40, 52, 3), dtype=np.uint8) * 255  # 0 / 255 only: the truncation edge
    out["sat"] = sat
    return out


def main():
    assert orc.cimg_available(), "needs oracle/_ref/libref_cimg.so (build container)"
    arrs = {}
    for name, img in inputs().items():
        n_levels = 1
        while min(img.shape[0], img.shape[1]) >> n_levels >= 1 and n_levels < 6:
            n_levels += 1
        levels = orc.cimg_pyramid(img, n_levels)
        assert np.array_equal(levels[0], img)
        arrs[f"{name}_in"] = img
        for l in range(1, n_levels):
            arrs[f"{name}_l{l}"] = levels[l]
    np.savez_compressed(os.path.join(HERE, "g4_cimg_pyramid.npz"), **arrs)
    print("wrote g4_cimg_pyramid.npz:", {k: v.shape for k, v in arrs.items()})


if __name__ == "__main__":
    main()
