"""Deterministic synthetic multi-view scene generator.

Bench/test input infrastructure, not part of the hot path.  The same arrays feed the
HIP path and the CPU oracle, so nothing here has to match the reference bit for bit;
what it has to do is exercise the path the way `Scene::initPatches`
(reference src/hpmvs/Scene.cpp:112-167) does: seeds near a textured surface, a first
image list from visibility, normal towards the first camera, scale from
`Camera::getScale(center, START_LEVEL)`.

Geometry : height field z = 0.3 sin(0.7 x) cos(0.9 y).
Texture  : per channel a sum of `n_waves` planar sinusoids over (x, y), wavelengths
           log-uniform in [lambda_min, 2] world units, mapped to uint8 [16, 240].
Cameras  : two rings (elevation 35 / 55 deg), radius 30, look-at origin, f = 1.2 W,
           NVM convention (f, quaternion wxyz world->camera, centre; reference
           include/hpmvs/NVMReader.h:44-50), k1 = 0.
Seeds    : Halton(2,3) points on the surface, displaced along the ray to the first
           visible camera by U(-1,1) * 0.5 * scale.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

SEED = 0x48504D56  # "HPMV"


def height(x, y, xp=np):
    return 0.3 * xp.sin(0.7 * x) * xp.cos(0.9 * y)


def height_grad(x, y, xp=np):
    return (0.3 * 0.7 * xp.cos(0.7 * x) * xp.cos(0.9 * y),
            -0.3 * 0.9 * xp.sin(0.7 * x) * xp.sin(0.9 * y))


def _quat_from_R(R: np.ndarray) -> np.ndarray:
    """Rotation matrix (world->camera) to quaternion wxyz, double precision."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    q = np.array([w, x, y, z], dtype=np.float64)
    return q / np.linalg.norm(q)


def _R_from_quat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


@dataclass
class View:
    width: int
    height: int
    f: float
    q: np.ndarray  # wxyz, world->camera
    c: np.ndarray  # centre (world)
    rgb: object = None  # uint8 [H, W, 3] (numpy, or a torch tensor when rendered on a device)


@dataclass
class Seeds:
    center: np.ndarray   # [N,4] float32, w = 1
    normal: np.ndarray   # [N,4] float32, w = 0
    scale: np.ndarray    # [N] float32
    n_images: np.ndarray  # [N] int32
    images: np.ndarray   # [N, max_images] int32 (unused slots -1)
    truth: np.ndarray    # [N,3] float64 surface point the seed was derived from


@dataclass
class SynthScene:
    views: list
    covis: list
    max_level: int = 5
    texture: dict = field(default_factory=dict)

    @property
    def n_views(self):
        return len(self.views)


def make_cameras(n_views: int, width: int, height: int, radius: float = 30.0):
    views = []
    for i in range(n_views):
        ring = i % 2
        elev = math.radians(35.0 if ring == 0 else 55.0)
        # neighbours at most 18 deg apart in azimuth so that small view counts stay covisible
        az = i * min(2.0 * math.pi / n_views, math.radians(18.0))
        c = np.array([radius * math.cos(elev) * math.cos(az), radius * math.cos(elev) * math.sin(az),
                      radius * math.sin(elev)], dtype=np.float64)
        fwd = -c / np.linalg.norm(c)
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        views.append(View(width, height, 1.2 * width, _quat_from_R(R), c))
    return views


def make_texture(n_waves: int, lambda_min: float, seed: int = SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    lam = np.exp(rng.uniform(math.log(lambda_min), math.log(2.0), size=(3, n_waves)))
    ang = rng.uniform(0, 2 * math.pi, size=(3, n_waves))
    phase = rng.uniform(0, 2 * math.pi, size=(3, n_waves))
    amp = rng.uniform(0.5, 1.0, size=(3, n_waves))
    kx = 2 * math.pi / lam * np.cos(ang)
    ky = 2 * math.pi / lam * np.sin(ang)
    norm = np.sqrt((amp ** 2).sum(axis=1) / 2.0)  # std of the sum
    return dict(kx=kx, ky=ky, phase=phase, amp=amp, norm=norm)


def _render_numpy(view: View, tex: dict, rows_per_chunk: int = 64) -> np.ndarray:
    W, H, f = view.width, view.height, view.f
    R = _R_from_quat(view.q)
    out = np.empty((H, W, 3), dtype=np.uint8)
    i = np.arange(W, dtype=np.float64)
    for y0 in range(0, H, rows_per_chunk):
        y1 = min(H, y0 + rows_per_chunk)
        j = np.arange(y0, y1, dtype=np.float64)
        dc = np.stack(np.broadcast_arrays((i[None, :] - W / 2.0) / f, (j[:, None] - H / 2.0) / f,
                                          np.ones((y1 - y0, W))), axis=-1)
        d = dc @ R  # camera->world = R^T applied to row vectors
        t = -view.c[2] / d[..., 2]
        for _ in range(10):
            px = view.c[0] + t * d[..., 0]
            py = view.c[1] + t * d[..., 1]
            t = (height(px, py) - view.c[2]) / d[..., 2]
        px = view.c[0] + t * d[..., 0]
        py = view.c[1] + t * d[..., 1]
        for ch in range(3):
            arg = px[..., None] * tex["kx"][ch] + py[..., None] * tex["ky"][ch] + tex["phase"][ch]
            v = (np.sin(arg) * tex["amp"][ch]).sum(axis=-1) / tex["norm"][ch]
            out[y0:y1, :, ch] = np.clip(128.0 + 45.0 * v, 16, 240).astype(np.uint8)
    return out


def _render_torch(view: View, tex: dict, device, rows_per_chunk: int = 256):
    import torch
    W, H, f = view.width, view.height, view.f
    dt = torch.float32
    R = torch.tensor(_R_from_quat(view.q), dtype=dt, device=device)
    c = [float(v) for v in view.c]
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
    i = torch.arange(W, dtype=dt, device=device)
    kx = torch.tensor(tex["kx"], dtype=dt, device=device)
    ky = torch.tensor(tex["ky"], dtype=dt, device=device)
    ph = torch.tensor(tex["phase"], dtype=dt, device=device)
    amp = torch.tensor(tex["amp"], dtype=dt, device=device)
    nrm = [float(v) for v in tex["norm"]]
    for y0 in range(0, H, rows_per_chunk):
        y1 = min(H, y0 + rows_per_chunk)
        j = torch.arange(y0, y1, dtype=dt, device=device)
        dcx = ((i - W / 2.0) / f)[None, :].expand(y1 - y0, W)
        dcy = ((j - H / 2.0) / f)[:, None].expand(y1 - y0, W)
        d0 = dcx * R[0, 0] + dcy * R[1, 0] + R[2, 0]
        d1 = dcx * R[0, 1] + dcy * R[1, 1] + R[2, 1]
        d2 = dcx * R[0, 2] + dcy * R[1, 2] + R[2, 2]
        t = -c[2] / d2
        for _ in range(10):
            px = c[0] + t * d0
            py = c[1] + t * d1
            t = (height(px, py, torch) - c[2]) / d2
        px = c[0] + t * d0
        py = c[1] + t * d1
        for ch in range(3):
            arg = px[..., None] * kx[ch] + py[..., None] * ky[ch] + ph[ch]
            v = (torch.sin(arg) * amp[ch]).sum(dim=-1) / nrm[ch]
            out[y0:y1, :, ch] = torch.clamp(128.0 + 45.0 * v, 16, 240).to(torch.uint8)
    return out


def make_scene(n_views: int, width: int, height: int, n_waves: int = 32, device=None,
               seed: int = SEED, max_level: int = 5) -> SynthScene:
    """Build cameras + rendered level-0 views.  `device` (a torch device) renders on the GPU."""
    views = make_cameras(n_views, width, height)
    px0 = 30.0 / (1.2 * width)           # world size of one level-0 pixel at the scene centre
    tex = make_texture(n_waves, lambda_min=max(0.02, 3.0 * px0), seed=seed)
    for v in views:
        v.rgb = _render_torch(v, tex, device) if device is not None else _render_numpy(v, tex)
    # covisibility: optical axes within 75 deg; lists in view order like Scene.cpp:268-276
    axes = [(_R_from_quat(v.q)[2]) for v in views]
    covis = []
    for a in range(n_views):
        covis.append([b for b in range(n_views)
                      if b != a and float(np.dot(axes[a], axes[b])) > math.cos(math.radians(75.0))])
    return SynthScene(views=views, covis=covis, max_level=max_level, texture=tex)


def _halton(n: int, base: int) -> np.ndarray:
    out = np.zeros(n)
    for i in range(n):
        f, r, k = 1.0, 0.0, i + 1
        while k > 0:
            f /= base
            r += f * (k % base)
            k //= base
        out[i] = r
    return out


def _halton_vec(n: int, base: int) -> np.ndarray:
    idx = np.arange(1, n + 1, dtype=np.int64)
    out = np.zeros(n)
    f = 1.0
    k = idx.copy()
    while k.max() > 0:
        f /= base
        out += f * (k % base)
        k //= base
    return out


def _make_seeds_torch(scene, n, start_level, max_images, max_seed_images, extent, seed, displace, device) -> Seeds:
    """make_seeds with the per-view visibility loop on a torch device (float64, the same formulas in the same
    order): BASELINE configs[4]'s 1e7 seeds against 200 views are minutes of numpy temporaries on the host and
    seconds here.  Device trigonometry may differ from numpy's in the last bit, so the two generators are two
    workloads of the same distribution, not the same arrays; whoever compares GPU and oracle hands BOTH the arrays
    this returns."""
    import torch
    f64 = torch.float64
    rng = np.random.Generator(np.random.PCG64(seed + 1))

    def halton(base):
        k = torch.arange(1, n + 1, dtype=torch.int64, device=device)
        out = torch.zeros(n, dtype=f64, device=device)
        f = 1.0
        while int(k.max()) > 0:
            f /= base
            out += f * (k % base).to(f64)
            k = k // base
        return out

    x = (halton(2) * 2 - 1) * extent
    y = (halton(3) * 2 - 1) * extent
    z = height(x, y, xp=torch)
    P = torch.stack([x, y, z], dim=1)
    gx, gy = height_grad(x, y, xp=torch)
    nrm = torch.stack([-gx, -gy, torch.ones_like(gx)], dim=1)
    nrm = nrm / torch.linalg.norm(nrm, dim=1, keepdim=True)
    Cn = np.stack([v.c for v in scene.views])
    C = torch.from_numpy(Cn).to(device)
    Rm = torch.from_numpy(np.stack([_R_from_quat(v.q) for v in scene.views])).to(device)
    W = np.array([v.width for v in scene.views]); H = np.array([v.height for v in scene.views])
    F = np.array([v.f for v in scene.views])
    lw = W // (1 << start_level); lh = H // (1 << start_level)
    images = -torch.ones((n, max_images), dtype=torch.int32, device=device)
    n_images = torch.zeros(n, dtype=torch.int64, device=device)
    margin = 2
    cos60 = math.cos(math.radians(60.0))
    for vi in range(scene.n_views):
        d = P - C[vi]
        pc = d @ Rm[vi].T
        u = (float(F[vi]) * pc[:, 0] / pc[:, 2] + float(W[vi]) / 2.0) / (1 << start_level)
        v = (float(F[vi]) * pc[:, 1] / pc[:, 2] + float(H[vi]) / 2.0) / (1 << start_level)
        ray = -d / torch.linalg.norm(d, dim=1, keepdim=True)
        vis = (pc[:, 2] > 0) & (u >= margin) & (v >= margin) & (u < int(lw[vi]) - margin) & (v < int(lh[vi]) - margin)
        vis &= (ray * nrm).sum(dim=1) > cos60
        vis &= n_images < max_seed_images
        rows = torch.nonzero(vis)[:, 0]
        images[rows, n_images[rows]] = vi
        n_images[rows] += 1
    first = torch.where(n_images > 0, images[:, 0].to(torch.int64), torch.zeros_like(n_images))
    Ft = torch.from_numpy(F.astype(np.float64)).to(device)
    c0 = C[first]
    to_cam = c0 - P
    dist = torch.linalg.norm(to_cam, dim=1)
    ray0 = to_cam / dist[:, None]
    scale0 = 2.0 * dist * (1 << start_level) / (2.0 * Ft[first])
    uni = torch.from_numpy(rng.uniform(-1, 1, size=n)).to(device)
    center = P + ray0 * (uni * displace * scale0)[:, None]
    one = torch.ones((n, 1), dtype=f64, device=device)
    c32 = torch.cat([center, one], dim=1).to(torch.float32)
    cc = torch.cat([c0, one], dim=1).to(torch.float32)
    nvec = cc - c32
    nvec = nvec / torch.sqrt((nvec * nvec).sum(dim=1))[:, None]
    nvec[:, 3] = 0.0
    fz = torch.sqrt(((c32 - cc) ** 2).sum(dim=1))
    scale = (2.0 * fz.to(f64) * (1 << start_level) / (2.0 * Ft[first].to(torch.float32).to(f64))).to(torch.float32)
    return Seeds(center=c32.cpu().numpy(), normal=nvec.cpu().numpy(), scale=scale.cpu().numpy(),
                 n_images=n_images.to(torch.int32).cpu().numpy(), images=images.cpu().numpy(), truth=P.cpu().numpy())


def make_seeds(scene: SynthScene, n: int, start_level: int = 4, max_images: int = 32, max_seed_images: int = 8,
               extent: float = 8.0, seed: int = SEED, displace: float = 0.5, device=None) -> Seeds:
    """Seed patches the way Scene::initPatches builds them (reference Scene.cpp:123-163).  `device` (a torch device):
    the same construction with torch on that device (see _make_seeds_torch)."""
    if device is not None:
        return _make_seeds_torch(scene, n, start_level, max_images, max_seed_images, extent, seed, displace, device)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = (_halton_vec(n, 2) * 2 - 1) * extent
    y = (_halton_vec(n, 3) * 2 - 1) * extent
    z = height(x, y)
    P = np.stack([x, y, z], axis=1)
    gx, gy = height_grad(x, y)
    nrm = np.stack([-gx, -gy, np.ones_like(gx)], axis=1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)

    nv = scene.n_views
    C = np.stack([v.c for v in scene.views])           # [V,3]
    Rm = np.stack([_R_from_quat(v.q) for v in scene.views])  # [V,3,3]
    W = np.array([v.width for v in scene.views]); H = np.array([v.height for v in scene.views])
    F = np.array([v.f for v in scene.views])
    lw = W // (1 << start_level); lh = H // (1 << start_level)

    images = -np.ones((n, max_images), dtype=np.int32)
    n_images = np.zeros(n, dtype=np.int32)
    margin = 2
    for vi in range(nv):
        d = P - C[vi]
        pc = d @ Rm[vi].T
        u = (F[vi] * pc[:, 0] / pc[:, 2] + W[vi] / 2.0) / (1 << start_level)
        v = (F[vi] * pc[:, 1] / pc[:, 2] + H[vi] / 2.0) / (1 << start_level)
        ray = -d / np.linalg.norm(d, axis=1, keepdims=True)
        vis = (pc[:, 2] > 0) & (u >= margin) & (v >= margin) & (u < lw[vi] - margin) & (v < lh[vi] - margin)
        vis &= (ray * nrm).sum(axis=1) > math.cos(math.radians(60.0))
        vis &= n_images < max_seed_images
        rows = np.nonzero(vis)[0]
        images[rows, n_images[rows]] = vi
        n_images[rows] += 1

    first = np.where(n_images > 0, images[:, 0], 0)
    c0 = C[first]
    to_cam = c0 - P
    dist = np.linalg.norm(to_cam, axis=1)
    ray0 = to_cam / dist[:, None]
    ffsum = 2.0 * F[first]
    scale0 = 2.0 * dist * (1 << start_level) / ffsum
    center = P + ray0 * (rng.uniform(-1, 1, size=n) * displace * scale0)[:, None]
    c32 = np.concatenate([center, np.ones((n, 1))], axis=1).astype(np.float32)
    # normal_/scale_3dx_ as Scene.cpp:158-163 computes them (float32, first attached camera)
    cc = np.concatenate([c0, np.ones((n, 1))], axis=1).astype(np.float32)
    nvec = (cc - c32).astype(np.float32)
    nvec /= np.sqrt((nvec * nvec).sum(axis=1, dtype=np.float32))[:, None]
    nvec[:, 3] = 0.0
    fz = np.sqrt(((c32 - cc) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32)
    scale = (2.0 * fz.astype(np.float64) * (1 << start_level) / (2.0 * F[first].astype(np.float32))).astype(np.float32)
    return Seeds(center=c32, normal=nvec.astype(np.float32), scale=scale, n_images=n_images,
                 images=images, truth=P)


def make_nvm_points(scene: SynthScene, n: int, start_level: int = 4, extent: float = 8.0, seed: int = SEED,
                    noise: float = 0.3):
    """NVM-style sparse points (reference NVM_Point: xyz + measurements[].imgIndex,
    include/hpmvs/NVMReader.h:38-42): surface points displaced by `noise` * (one START_LEVEL sample
    step) along the viewing ray, each measured in every camera that sees it (ascending index,
    within 60 deg of the surface normal).  Returns (xyz[n,3] float64, meas_off[n+1], meas_img)."""
    rng = np.random.Generator(np.random.PCG64(seed + 3))
    x = (_halton_vec(n, 2) * 2 - 1) * extent
    y = (_halton_vec(n, 3) * 2 - 1) * extent
    P = np.stack([x, y, height(x, y)], axis=1)
    gx, gy = height_grad(x, y)
    nrm = np.stack([-gx, -gy, np.ones_like(gx)], axis=1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    C = np.stack([v.c for v in scene.views])
    Rm = np.stack([_R_from_quat(v.q) for v in scene.views])
    vis = np.zeros((n, scene.n_views), dtype=bool)
    for vi, v in enumerate(scene.views):
        d = P - C[vi]
        pc = d @ Rm[vi].T
        u = v.f * pc[:, 0] / pc[:, 2] + v.width / 2.0
        w = v.f * pc[:, 1] / pc[:, 2] + v.height / 2.0
        ray = -d / np.linalg.norm(d, axis=1, keepdims=True)
        vis[:, vi] = (pc[:, 2] > 0) & (u >= 0) & (w >= 0) & (u < v.width) & (w < v.height) & \
                     ((ray * nrm).sum(axis=1) > math.cos(math.radians(60.0)))
    first = np.argmax(vis, axis=1)
    to_cam = C[first] - P
    dist = np.linalg.norm(to_cam, axis=1)
    step = 2.0 * dist * (1 << start_level) / (2.0 * np.array([scene.views[i].f for i in first]))
    xyz = P + (to_cam / dist[:, None]) * (rng.uniform(-1, 1, size=n) * noise * step)[:, None]
    counts = vis.sum(axis=1)
    meas_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    meas_img = np.nonzero(vis)[1].astype(np.int32)
    return xyz, meas_off, meas_img
