"""Patch-index sharding + the per-round all-gather of refined-patch records.

The refinement path has no data dependence between patches (reference
src/hpmvs/PatchOptimizer.cpp:78-103 reads only its own Patch3d and the immutable scene; the
reference itself runs it as an OpenMP loop over seeds, src/hpmvs/Scene.cpp:114).  One process per
GPU refines a contiguous block of the batch against its own replica of the scene; the only
exchange is ONE all-gather per expansion round of fixed-size records, so that every rank holds
the full refined set before the host scheduler inserts it (RCCL over xGMI on the GPU box, gloo in
the CPU tests).  No all-reduce, no ring dependency.

Record (192 bytes, the layout of `hpmvs_record` in include/hpmvs_amd.h -- the C ABI packs the same
bytes on the device with `hpmvs_pack_records`):

    0   center[4] f32      16  normal[4] f32     32  color[3] f32     44  scale f32
    48  fmin f64           56  ok u8             57  pad u8           58  n_images u16
    60  pad[4]             64  images[64] u16 (unused slots 0xFFFF)

All 64 image ids of the ABI (`HPMVS_MAX_IMAGES`) travel; an id that does not fit 16 bits or a list
longer than 64 raises instead of truncating.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 192
MAX_RECORD_IMAGES = 64  # == HPMVS_MAX_IMAGES; ids stored as uint16
NO_IMAGE = 0xFFFF


def shard_range(n: int, rank: int, world: int):
    """Contiguous block partition [lo, hi) of n patches."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def tile_order(center, images, n_images, cameras_P0, tile_px: int = 256) -> np.ndarray:
    """SURVEY 8e's host-side bucket sort: a permutation that groups the patches of a round by (reference image,
    256-px tile of the projected centre in that image), so that wavefronts working on neighbouring queue entries
    sample neighbouring pixels.  Patches are independent, so the order changes no result -- only which cache
    lines a wavefront's taps share.  cameras_P0: [n_views, 12] level-0 projection rows (hpmvs_camera.P0)."""
    center = np.asarray(center, dtype=np.float64)
    ref = np.where(np.asarray(n_images) > 0, np.asarray(images)[:, 0], 0).astype(np.int64)
    P = np.asarray(cameras_P0, dtype=np.float64).reshape(-1, 3, 4)[ref]
    X = np.concatenate([center[:, :3], np.ones((len(center), 1))], axis=1)
    r = np.einsum("nij,nj->ni", P, X)
    z = np.where(np.abs(r[:, 2]) > 1e-12, r[:, 2], 1.0)
    tx = np.clip(np.floor(r[:, 0] / z / tile_px), -1, 1 << 15).astype(np.int64)
    ty = np.clip(np.floor(r[:, 1] / z / tile_px), -1, 1 << 15).astype(np.int64)
    key = (ref << 40) | ((ty + 1) << 20) | (tx + 1)
    return np.argsort(key, kind="stable")


def pack_records(center, normal, color, scale, fmin, ok, n_images, images, check: bool = True) -> torch.Tensor:
    """SoA outputs -> [n, 192] uint8 records (same layout on every rank, any device).

    `check` validates the ranges on the device (one small reduction + sync); the timed path of
    bench.py validates once outside the timed region and passes check=False."""
    n = scale.shape[0]
    cols = images.shape[1]
    if cols > MAX_RECORD_IMAGES:
        raise ValueError(f"images has {cols} columns; the record carries {MAX_RECORD_IMAGES}")
    if check and n > 0:
        nmax = int(n_images.max())
        if nmax > cols or nmax > MAX_RECORD_IMAGES:
            raise ValueError(f"a patch has {nmax} images but only {min(cols, MAX_RECORD_IMAGES)} ids can travel")
        valid = torch.arange(cols, device=images.device)[None, :] < n_images.to(images.device)[:, None].clamp(min=0)
        live = images[valid]
        if live.numel() and (int(live.max()) >= NO_IMAGE or int(live.min()) < 0):
            raise ValueError("image id outside [0, 65534] cannot be stored in the uint16 record")
    rec = torch.zeros((n, RECORD_BYTES), dtype=torch.uint8, device=scale.device)
    rec[:, 0:16] = center.contiguous().view(torch.uint8).view(n, 16)
    rec[:, 16:32] = normal.contiguous().view(torch.uint8).view(n, 16)
    rec[:, 32:44] = color.contiguous().view(torch.uint8).view(n, 12)
    rec[:, 44:48] = scale.contiguous().view(torch.uint8).view(n, 4)
    rec[:, 48:56] = fmin.contiguous().view(torch.uint8).view(n, 8)
    rec[:, 56] = ok.to(torch.uint8)
    rec[:, 58:60] = n_images.to(torch.int16).contiguous().view(torch.uint8).view(n, 2)
    ids = torch.full((n, MAX_RECORD_IMAGES), -1, dtype=torch.int16, device=scale.device)  # 0xFFFF
    nim = n_images.to(images.device)[:, None].clamp(min=0)
    col = torch.arange(cols, device=images.device)[None, :]
    # ids >= 32768 wrap to the same 16 bits an unsigned store would write
    ids[:, :cols] = torch.where(col < nim, images, torch.full_like(images, -1)).to(torch.int16)
    rec[:, 64:192] = ids.view(torch.uint8).view(n, 128)
    return rec


def unpack_records(rec: torch.Tensor) -> dict:
    rec = rec.cpu().contiguous()
    n = rec.shape[0]
    f32 = lambda a, b, c: rec[:, a:b].contiguous().view(torch.float32).view(n, c).numpy()
    ids = rec[:, 64:192].contiguous().view(torch.int16).view(n, MAX_RECORD_IMAGES).numpy().astype(np.uint16)
    images = ids.astype(np.int32)
    images[ids == NO_IMAGE] = -1
    return dict(center=f32(0, 16, 4), normal=f32(16, 32, 4), color=f32(32, 44, 3), scale=f32(44, 48, 1)[:, 0],
                fmin=rec[:, 48:56].contiguous().view(torch.float64).view(n).numpy(),
                ok=rec[:, 56].numpy().astype(bool),
                # signed like the C ABI's unpack_records_kernel: seed_init's rejection codes are negative counts
                n_images=rec[:, 58:60].contiguous().view(torch.int16).view(n).numpy().astype(np.int32),
                images=images)


def all_gather_records(local: torch.Tensor, counts=None) -> torch.Tensor:
    """One collective per round.  Equal shard sizes use all_gather_into_tensor; ragged shards
    (n not divisible by the world size) are padded to the largest shard and trimmed."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()  # a 1-rank group still goes through the collective (bench.py's forced-dist test)
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * local.shape[0], RECORD_BYTES), dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    m = max(counts)
    padded = torch.zeros((m, RECORD_BYTES), dtype=torch.uint8, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * m, RECORD_BYTES), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)], dim=0)
