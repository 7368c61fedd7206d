"""BASELINE.json configs[0] ("tiny nvm scene") end to end through real files: an NVM_V3 model + binary PPM
views -> C++ NVMReader -> Scene::addCameras / extractCoVisiblilty / initPatches (GPU) -> extended PLY.
Field-wise parity: the NVM round trip reproduces cameras and points, and the PLY holds exactly the patches
the C ABI returns for the same points."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_nvm(path, scene, xyz, off, img):
    with open(path, "w") as f:
        f.write("NVM_V3\n\n%d\n" % scene.n_views)
        for i, v in enumerate(scene.views):
            f.write("view%02d.ppm %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g 0 0\n" %
                    (i, v.f, v.q[0], v.q[1], v.q[2], v.q[3], v.c[0], v.c[1], v.c[2]))
        f.write("\n%d\n" % len(xyz))
        for k in range(len(xyz)):
            ms = img[off[k]:off[k + 1]]
            f.write("%.17g %.17g %.17g 128 128 128 %d" % (xyz[k, 0], xyz[k, 1], xyz[k, 2], len(ms)))
            for j, m in enumerate(ms):
                f.write(" %d %d 0 0" % (m, k))
            f.write("\n")
        f.write("\n0\n")


def read_nvm(path):
    tok = open(path).read().split()
    assert tok[0] == "NVM_V3"
    p = 1
    nc = int(tok[p]); p += 1
    cams = []
    for _ in range(nc):
        cams.append((tok[p], [float(t) for t in tok[p + 1:p + 10]])); p += 11
    npts = int(tok[p]); p += 1
    pts = []
    for _ in range(npts):
        xyz = [float(t) for t in tok[p:p + 3]]; nm = int(tok[p + 6]); p += 7
        ms = [int(tok[p + 4 * j]) for j in range(nm)]; p += 4 * nm
        pts.append((xyz, ms))
    return cams, pts


def read_ext_ply(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.decode().splitlines() if l.startswith("element vertex")][0].split()[2])
    assert b"binary_little_endian" in head and b"scalar_scale" in head and b"visible_cameras" in head
    rec = np.dtype([("xyz", "<f4", 3), ("n", "<f4", 3), ("rgb", "u1", 3), ("scale", "<f4")])
    v = np.frombuffer(body[: n * rec.itemsize], dtype=rec)
    p = n * rec.itemsize
    vis = []
    for _ in range(n):
        k = struct.unpack_from("<I", body, p)[0]; p += 4
        vis.append(list(struct.unpack_from("<%dI" % k, body, p))); p += 4 * k
    assert p == len(body)
    return v, vis


def test_tiny_nvm_scene_through_files(tiny_scene, gpu_scene, tmp_path):
    from hpmvs_amd import api, synth
    exe = os.path.join(ROOT, "tests", "native", "run_nvm_scene")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe), "all"], check=True, capture_output=True)
    xyz, off, img = synth.make_nvm_points(tiny_scene, 300, start_level=2, noise=1.0)
    for i, v in enumerate(tiny_scene.views):
        with open(tmp_path / ("view%02d.ppm" % i), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (v.width, v.height) + np.ascontiguousarray(v.rgb).tobytes())
    nvm = tmp_path / "scene.nvm"
    write_nvm(nvm, tiny_scene, xyz, off, img)
    out = subprocess.run([exe, str(nvm), str(tmp_path / "out.ply"), str(tmp_path / "copy.nvm"), "2"], check=True,
                         capture_output=True, text=True).stdout
    # NVM round trip (saveNVM writes 12 significant digits like the reference)
    cams, pts = read_nvm(tmp_path / "copy.nvm")
    assert len(cams) == tiny_scene.n_views and len(pts) == len(xyz)
    for (name, par), v in zip(cams, tiny_scene.views):
        assert name.endswith(".ppm") and np.allclose(par[:8], [v.f, *v.q, *v.c], rtol=1e-11, atol=1e-11)
    for k, (p, ms) in enumerate(pts):
        assert np.allclose(p, xyz[k], rtol=1e-11, atol=1e-11) and ms == list(img[off[k]:off[k + 1]])
    # the PLY holds exactly what the C ABI returns for the same points
    batch = api.init_patches_batch(gpu_scene, xyz, off, img, start_level=2, max_images=64)
    good = np.nonzero(batch.ok)[0]
    v, vis = read_ext_ply(tmp_path / "out.ply")
    assert "refined %d " % len(good) in out and len(v) == len(good) and len(good) >= 50
    assert np.array_equal(v["xyz"], batch.center[good, :3]) and np.array_equal(v["n"], batch.normal[good, :3])
    assert np.array_equal(v["scale"], batch.scale[good])
    assert np.array_equal(v["rgb"], batch.color[good].astype(np.uint8))
    for r, k in enumerate(good):
        assert vis[r] == list(batch.images[k, :batch.n_images[k]])
