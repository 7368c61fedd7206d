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

The record carries 64 ids (`HPMVS_RECORD_IMAGES`).  A refined list longer than that (the ABI takes up to
`HPMVS_MAX_IMAGES` = 256; rare: a dense scene) keeps its true count and its first 64 ids in the record and sends the
rest as a TAIL (`hpmvs_record_tail`, 392 bytes: patch index i32, count u16, pad u16, 192 ids u16) in a second
all-gather that is made only when some rank has a tail -- every rank sees that in the gathered records
(`all_gather_round`).  An id that does not fit 16 bits, or a list longer than its row, raises instead of truncating.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 192
MAX_RECORD_IMAGES = 64  # == HPMVS_RECORD_IMAGES; ids stored as uint16
MAX_LIST_IMAGES = 256   # == HPMVS_MAX_IMAGES
TAIL_BYTES = 392        # hpmvs_record_tail
TAIL_IDS = MAX_LIST_IMAGES - MAX_RECORD_IMAGES
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
    if cols > MAX_LIST_IMAGES:
        raise ValueError(f"images has {cols} columns; the ABI takes {MAX_LIST_IMAGES}")
    if check and n > 0:
        nmax = int(n_images.max())
        if nmax > cols:
            raise ValueError(f"a patch has {nmax} images but its row holds {cols} ids")
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
    rc = min(cols, MAX_RECORD_IMAGES)   # (ids from 64 on travel in tails: pack_tails)
    ids[:, :rc] = torch.where(col < nim, images, torch.full_like(images, -1)).to(torch.int16)[:, :rc]
    rec[:, 64:192] = ids.view(torch.uint8).view(n, 128)
    return rec


def pack_tails(ok, n_images, images) -> torch.Tensor:
    """[t, 392] uint8: the tails of the refined patches (ok != 0) whose list is longer than the record's 64 ids, in patch
    order -- the bytes `hpmvs_pack_record_tails` writes."""
    n, cols = images.shape
    dev = images.device
    if cols <= MAX_RECORD_IMAGES or n == 0:
        return torch.zeros((0, TAIL_BYTES), dtype=torch.uint8, device=dev)
    nim = n_images.to(dev).to(torch.int64)
    has = (ok.to(dev) != 0) & (nim > MAX_RECORD_IMAGES) & (nim <= cols)
    idx = torch.nonzero(has)[:, 0]
    t = int(idx.numel())
    out = torch.zeros((t, TAIL_BYTES), dtype=torch.uint8, device=dev)
    if t == 0:
        return out
    out[:, 0:4] = idx.to(torch.int32).contiguous().view(torch.uint8).view(t, 4)
    cnt = (nim[idx] - MAX_RECORD_IMAGES)
    out[:, 4:6] = cnt.to(torch.int16).contiguous().view(torch.uint8).view(t, 2)
    ids = torch.full((t, TAIL_IDS), -1, dtype=torch.int16, device=dev)
    w = min(cols - MAX_RECORD_IMAGES, TAIL_IDS)
    col = torch.arange(w, device=dev)[None, :]
    src = images[idx][:, MAX_RECORD_IMAGES:MAX_RECORD_IMAGES + w]
    ids[:, :w] = torch.where(col < cnt[:, None], src, torch.full_like(src, -1)).to(torch.int16)
    out[:, 8:8 + 2 * TAIL_IDS] = ids.view(torch.uint8).view(t, 2 * TAIL_IDS)
    return out


def all_gather_round(records: torch.Tensor, tails: torch.Tensor, counts=None):
    """The round's exchange: ONE all-gather of the 192-byte records, and -- only if the gathered records show that some rank
    has lists longer than 64 ids -- one more of the tails (padded to the largest rank's number).  Returns (records of the
    whole round, tails of the whole round with `patch` rebased to round indices)."""
    gathered = all_gather_records(records, counts)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if counts is None:
        counts = [records.shape[0]] * world
    g = gathered
    nim = g[:, 58:60].contiguous().view(torch.int16).view(-1).to(torch.int64)
    long = (g[:, 56] != 0) & (nim > MAX_RECORD_IMAGES)
    per_rank, lo = [], 0
    for c in counts:
        per_rank.append(int(long[lo:lo + c].sum()))
        lo += c
    if sum(per_rank) == 0:
        return gathered, torch.zeros((0, TAIL_BYTES), dtype=torch.uint8, device=records.device)
    if world == 1:
        return gathered, tails
    m = max(per_rank)
    padded = torch.zeros((m, TAIL_BYTES), dtype=torch.uint8, device=records.device)
    padded[: tails.shape[0]] = tails
    out = torch.empty((world * m, TAIL_BYTES), dtype=torch.uint8, device=records.device)
    dist.all_gather_into_tensor(out, padded)
    parts, lo = [], 0
    for r, c in enumerate(counts):
        part = out[r * m: r * m + per_rank[r]].clone()
        if part.shape[0]:
            p = part[:, 0:4].contiguous().view(torch.int32).view(-1) + lo   # shard index -> round index
            part[:, 0:4] = p.contiguous().view(torch.uint8).view(-1, 4)
        parts.append(part)
        lo += c
    return gathered, torch.cat(parts, dim=0)


def unpack_records(rec: torch.Tensor, tails: torch.Tensor = None) -> dict:
    """Records (and, if any, the round's tails with round patch indices) -> arrays; `images` is 64 columns wide, or as wide
    as the longest list when tails are given."""
    rec = rec.cpu().contiguous()
    n = rec.shape[0]
    f32 = lambda a, b, c: rec[:, a:b].contiguous().view(torch.float32).view(n, c).numpy()
    ids = rec[:, 64:192].contiguous().view(torch.int16).view(n, MAX_RECORD_IMAGES).numpy().astype(np.uint16)
    images = ids.astype(np.int32)
    images[ids == NO_IMAGE] = -1
    if tails is not None and tails.shape[0]:
        tl = tails.cpu().contiguous()
        t = tl.shape[0]
        patch = tl[:, 0:4].contiguous().view(torch.int32).view(t).numpy()
        cnt = tl[:, 4:6].contiguous().view(torch.int16).view(t).numpy().astype(np.int64)
        tid = tl[:, 8:8 + 2 * TAIL_IDS].contiguous().view(torch.int16).view(t, TAIL_IDS).numpy().astype(np.uint16)
        wide = np.full((n, MAX_RECORD_IMAGES + int(cnt.max())), -1, np.int32)
        wide[:, :MAX_RECORD_IMAGES] = images
        for k in range(t):
            wide[patch[k], MAX_RECORD_IMAGES:MAX_RECORD_IMAGES + cnt[k]] = tid[k, :cnt[k]].astype(np.int32)
        images = wide
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
